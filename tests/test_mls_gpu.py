"""pgp_mls_normals (csrc/mls.hip) through the C ABI: against the C restatement (same steps, neighbours
visited in another order: float outputs of double sums) and against the independent numpy fixture."""
import os

import numpy as np
import pytest

from physimglobalpose_amd import LcpScorer, synth
from _checkers import oracle_mls
from test_mls_oracle import check_against_fixture, GOLD

pytestmark = pytest.mark.gpu


def agree(got, want):
    ox, on, oc, oi = got
    wx, wn, wc, wi = want
    assert np.array_equal(oi, wi)
    assert np.abs(ox - wx).max() <= 1.5e-7          # ~2 ulp at 0.6 m
    assert np.abs(on - wn).max() <= 2e-6            # same sign: both follow pcl::eigen33's cross products
    assert np.allclose(oc, wc, rtol=1e-5, atol=1e-10)


def test_fixture_and_restatement():
    g = np.load(GOLD)
    sc = LcpScorer(0)
    got = sc.mls_normals(g["xyz"], float(g["radius"]))
    check_against_fixture(got, g)
    agree(got, oracle_mls(g["xyz"], float(g["radius"])))


def test_segment_sized_cloud_from_the_voxel_grid():
    """the reference's chain: back-projected segment -> 1 cm voxel grid -> MLS (Segmentation.cpp:234-246)"""
    w = synth.make_workload(20000, 1000, 4, config_id=3)
    sc = LcpScorer(0)
    seg = sc.voxel_grid(w.P_xyz[:8000], 0.01)
    assert len(seg) > 500
    got = sc.mls_normals(seg, 0.02)
    agree(got, oracle_mls(seg, 0.02))
    assert len(got[3]) > 0.5 * len(seg)


def test_edges():
    sc = LcpScorer(0)
    assert len(sc.mls_normals(np.zeros((0, 3), np.float32))[3]) == 0
    two = np.array([[0, 0, 0.5], [0.005, 0, 0.5]], np.float32)
    assert len(sc.mls_normals(two)[3]) == 0
    pts = np.array([[0, 0, 0.5], [0.005, 0, 0.5], [0, 0.005, 0.5], [np.nan, 0, 0], [5, 5, 5]], np.float32)
    ox, on, oc, oi = sc.mls_normals(pts)
    assert list(oi) == [0, 1, 2]                    # the non-finite and the isolated point are dropped
    agree((ox, on, oc, oi), oracle_mls(pts))


def test_device_pointer_form():
    import torch
    g = np.load(GOLD)
    sc = LcpScorer(0)
    n = len(g["xyz"])
    d = torch.from_numpy(g["xyz"]).cuda()
    ox, on = torch.zeros(n, 3, device="cuda"), torch.zeros(n, 3, device="cuda")
    oc, oi = torch.zeros(n, device="cuda"), torch.zeros(n, dtype=torch.int32, device="cuda")
    m = sc.mls_normals_device(d, n, float(g["radius"]), ox, on, oc, oi)
    host = sc.mls_normals(g["xyz"], float(g["radius"]))
    assert m == len(host[3])
    assert np.array_equal(ox[:m].cpu().numpy(), host[0]) and np.array_equal(on[:m].cpu().numpy(), host[1])
    assert np.array_equal(oc[:m].cpu().numpy(), host[2]) and np.array_equal(oi[:m].cpu().numpy(), host[3])


def test_degenerate_neighbourhoods_follow_the_restatement():
    """collinear points, an exact planar lattice, coincident points: whatever PCL's closed forms make of a
    rank-deficient covariance (zero roots, zero-length cross products -> NaN), the kernel makes the same"""
    sc = LcpScorer(0)
    line = np.stack([np.arange(12) * 0.004, np.zeros(12), np.full(12, 0.5)], 1).astype(np.float32)
    lattice = np.stack(np.meshgrid(np.arange(8) * 0.005, np.arange(8) * 0.005, [0.7], indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    same = np.tile(np.array([[0.1, 0.2, 0.6]], np.float32), (9, 1))
    for cloud in (line, lattice, same, np.vstack([line, lattice + np.float32(1.0)])):
        got, want = sc.mls_normals(cloud, 0.02), oracle_mls(cloud, 0.02)
        assert np.array_equal(got[3], want[3])
        for g, w in zip(got[:3], want[:3]):
            assert np.array_equal(np.isnan(g), np.isnan(w))
            ok = ~np.isnan(w)
            assert np.allclose(g[ok], w[ok], rtol=1e-5, atol=2e-6)
