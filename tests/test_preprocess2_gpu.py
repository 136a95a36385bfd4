"""Voxel grid (Segmentation.cpp:234-237), Hausdorff pose distances (base.cc:1616-1655) and the
device-resident chain depth image -> cloud -> voxel grid -> scene index -> scores."""
import os

import numpy as np
import pytest

from physimglobalpose_amd import LcpScorer, PGP_MODE_PLAIN, synth
from _checkers import oracle_pose_hausdorff, oracle_voxel_grid

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def numpy_voxel_grid(xyz, leaf):
    """A third, independent statement of the same rule (float32 arithmetic, per-voxel cumulative sums)."""
    xyz = xyz[np.isfinite(xyz).all(1)]
    inv = np.float32(1.0) / np.float32(leaf)
    min_b = np.floor(xyz.min(0) * inv).astype(np.int64)
    div = np.floor(xyz.max(0) * inv).astype(np.int64) - min_b + 1
    ijk = (np.floor(xyz * inv) - min_b.astype(np.float32)).astype(np.int64)
    key = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    order = np.argsort(key, kind="stable")
    out = []
    for run in np.split(order, np.flatnonzero(np.diff(key[order])) + 1):
        c = np.zeros(3, np.float32)
        for p in xyz[run]:
            c = (c + p).astype(np.float32)
        out.append(c / np.float32(len(run)))
    return np.array(out, np.float32)


def test_voxel_grid_oracle_against_numpy_statement():
    rng = np.random.default_rng(1)
    xyz = rng.uniform(-0.2, 0.3, (5000, 3)).astype(np.float32)
    xyz[::97] = np.nan
    assert np.array_equal(oracle_voxel_grid(xyz, 0.01), numpy_voxel_grid(xyz, 0.01))


def test_hausdorff_oracle_matches_golden():
    g = np.load(os.path.join(GOLD, "hausdorff.npz"))
    dmax, dsum = oracle_pose_hausdorff(g["hull"], g["T"], g["pairs"])
    assert np.array_equal(dmax, g["dmax"]) and np.array_equal(dsum, g["dsum"])


@pytest.mark.gpu
def test_voxel_grid_bit_exact():
    sc = LcpScorer()
    rng = np.random.default_rng(2)
    for n, leaf in ((20000, 0.01), (3000, 0.005), (1, 0.01), (700, 0.25)):
        xyz = (rng.uniform(-0.4, 0.6, (n, 3)) * [1, 0.5, 0.2] + [0, 0, 0.8]).astype(np.float32)
        if n > 100:
            xyz[::53] = np.inf
        got = sc.voxel_grid(xyz, leaf)
        assert np.array_equal(got, oracle_voxel_grid(xyz, leaf)), (n, leaf)
    assert len(sc.voxel_grid(np.zeros((0, 3), np.float32))) == 0
    assert len(sc.voxel_grid(np.full((5, 3), np.nan, np.float32))) == 0


@pytest.mark.gpu
def test_voxel_grid_on_the_real_frame():
    fr = np.load(os.path.join(GOLD, "test_scene_frame.npz"))
    sg = np.load(os.path.join(GOLD, "test_scene_segments.npz"))
    sc = LcpScorer()
    for cls in (2, 3, 8):
        dense = sc.backproject_depth(fr["raw"], fr["K"], (fr["mask"] == cls).astype(np.uint8))
        leaves = sc.voxel_grid(dense, 0.01)
        assert np.array_equal(leaves, oracle_voxel_grid(dense, 0.01))
        # the segment fixture (first point of every 1 cm voxel, float64 keys, z in 0.2..2.0) has the same leaves
        assert abs(len(leaves) - len(sg[f"seg_{cls}"])) <= 0.03 * len(leaves)


@pytest.mark.gpu
def test_hausdorff_bit_exact():
    g = np.load(os.path.join(GOLD, "hausdorff.npz"))
    sc = LcpScorer()
    dmax, dsum = sc.pose_hausdorff(g["hull"], g["T"], g["pairs"])
    assert np.array_equal(dmax, g["dmax"]) and np.array_equal(dsum, g["dsum"])
    assert (dmax[:10] == 0).all()      # a pose against itself
    # ragged hull sizes around the 64-lane chunks, against the C restatement
    rng = np.random.default_rng(3)
    for nh in (1, 63, 64, 65, 130):
        hull = rng.uniform(-0.1, 0.1, (nh, 3)).astype(np.float32)
        a, b = sc.pose_hausdorff(hull, g["T"], g["pairs"][:50])
        oa, ob = oracle_pose_hausdorff(hull, g["T"], g["pairs"][:50])
        assert np.array_equal(a, oa) and np.array_equal(b, ob), nh


@pytest.mark.gpu
def test_device_resident_chain_equals_host_path():
    """depth image -> cloud -> 1 cm voxel grid -> scene index -> scores, once through host arrays and once
    with every intermediate left in HBM (the *_device entry points): identical scores; clustering of
    the scored poses from device arrays equals the host-pointer call."""
    import torch
    fr = np.load(os.path.join(GOLD, "test_scene_frame.npz"))
    raw, K, mask = fr["raw"], fr["K"], (fr["mask"] == 8).astype(np.uint8)
    rng = np.random.default_rng(4)
    host = LcpScorer(0)
    dense = host.backproject_depth(raw, K, mask)
    leaves = host.voxel_grid(dense, 0.01)
    model = (dense[rng.choice(len(dense), 2000, replace=False)] - dense.mean(0)).astype(np.float32)
    host.set_scene(leaves, None, None, 0.005)
    host.set_model(model)
    T = np.stack([synth.colmajor16(synth._se3(synth._random_rot(rng, np.deg2rad(3.0)),
                                              dense.mean(0) + 0.004 * rng.standard_normal(3))) for _ in range(512)])
    s_host, c_host, bi_host, bs_host = host.score(T, PGP_MODE_PLAIN)

    dev = LcpScorer(0)
    d_raw = torch.from_numpy(raw.view(np.int16)).cuda()
    d_mask = torch.from_numpy(mask).cuda()
    d_cloud = torch.zeros(raw.size, 3, device="cuda")
    n = dev.backproject_depth_device(d_raw, K, d_mask, d_cloud)
    assert n == len(dense) and np.array_equal(d_cloud[:n].cpu().numpy(), dense)
    d_leaves = torch.zeros(n, 3, device="cuda")
    m = dev.voxel_grid_device(d_cloud, n, 0.01, d_leaves)
    assert m == len(leaves) and np.array_equal(d_leaves[:m].cpu().numpy(), leaves)
    dev.set_scene_device(d_leaves, m, delta=0.005)
    dev.set_model(model)
    s_dev, c_dev, bi_dev, bs_dev = dev.score(T, PGP_MODE_PLAIN)
    assert np.array_equal(s_dev, s_host) and np.array_equal(c_dev, c_host) and (bi_dev, bs_dev) == (bi_host, bs_host)
    assert s_host.max() > 0.3

    rep, assign = host.cluster_poses(T, s_host + np.float32(1e-6), bs_host, accept_fraction=0.0)
    d_T, d_s = torch.from_numpy(T).cuda(), torch.from_numpy(s_host + np.float32(1e-6)).cuda()
    d_rep = torch.zeros(len(T), dtype=torch.int32, device="cuda")
    d_assign = torch.zeros(len(T), dtype=torch.int32, device="cuda")
    n_rep = dev.cluster_poses_device(d_T, d_s, bs_host, d_rep, d_assign, accept_fraction=0.0)
    assert n_rep == len(rep) and np.array_equal(d_rep[:n_rep].cpu().numpy(), rep)
    assert np.array_equal(d_assign.cpu().numpy(), assign)
