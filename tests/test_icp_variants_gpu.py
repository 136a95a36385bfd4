"""The ICP forms of the reference's call sites (csrc/icp.hip, pgp_icp_refine_ex) against fixtures from
an INDEPENDENT implementation (tests/golden/make_icp_golden.py: numpy float64, scipy cKDTree, SVD /
Kabsch, np.linalg.solve) -- PCL and libpointmatcher are neither vendored in the reference nor
installed, so agreement of two separately written codes is the pin this path can have.  Also: the
uniform-grid nearest-neighbour search gives the results of the exhaustive scan bit for bit."""
import ast
import os

import numpy as np
import pytest

from physimglobalpose_amd import LcpScorer, synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "icp.npz")
CASES = ["trimmed", "capped", "plain", "plane", "pointmatcher"]


def _pose_diff(A, B):
    D = np.linalg.inv(A) @ B
    ang = np.degrees(np.arccos(np.clip((np.trace(D[:3, :3]) - 1) / 2, -1, 1)))
    return ang, np.linalg.norm(A[:3, 3] - B[:3, 3])


@pytest.mark.parametrize("name", CASES)
def test_matches_the_independent_implementation(name):
    g = np.load(GOLD)
    src = g[str(g[f"{name}_src"][0])]
    opts = ast.literal_eval(str(g[f"{name}_opts"][0]))
    sc = LcpScorer()
    G0 = np.stack([synth.colmajor16(G) for G in g["guesses"]])
    T, energy, iters = sc.icp_refine_ex(src, g["model"], G0, tgt_nrm=g["normals"], **opts)
    want_it = g[f"{name}_it"]
    for k in range(len(T)):
        got = T[k].reshape(4, 4).T.astype(np.float64)
        ang, dt = _pose_diff(g[f"{name}_G"][k], got)
        settled = want_it[k] < opts["max_iterations"]
        if settled:
            # both codes stopped on their rule: same optimum (float32 vs float64 distances move the stopping
            # iteration by a step or two at most), same energy
            assert ang < 0.02 and dt < 5e-5, (name, k, ang, dt, iters[k], want_it[k])
            assert abs(int(iters[k]) - int(want_it[k])) <= 2, (name, k, iters[k], want_it[k])
            assert abs(energy[k] - g[f"{name}_E"][k]) <= 0.01 * g[f"{name}_E"][k]
        else:
            # iteration cap reached while still moving: the trajectories agree to the accumulated rounding
            assert int(iters[k]) == int(want_it[k]) and ang < 0.5 and dt < 2e-3, (name, k, ang, dt)


def test_grid_search_equals_exhaustive_scan():
    """Scene-sized capped ICP (SceneCfg.cpp:135-141 form): nn_search=2 (uniform grid) and nn_search=1
    (exhaustive scan) return identical transforms, energies and iteration counts."""
    rng = np.random.default_rng(11)
    w = synth.make_workload(30000, 3000, 2, config_id=141)
    tgt = w.P_xyz
    R = synth._random_rot(rng, np.deg2rad(1.5))
    src = (tgt[rng.choice(len(tgt), 20000, replace=False)] @ R.T + 0.003 * rng.standard_normal(3)
           + 0.0005 * rng.standard_normal((20000, 3))).astype(np.float32)
    G0 = np.stack([synth.colmajor16(synth._se3(synth._random_rot(rng, np.deg2rad(0.5)), 0.002 * rng.standard_normal(3)))
                   for _ in range(3)])
    sc = LcpScorer()
    kw = dict(max_iterations=12, max_corr_dist=0.01, energy_ratio=0.0, transformation_epsilon=1e-9, absolute_mse=1e-12)
    Ta, Ea, ia = sc.icp_refine_ex(src, tgt, G0, nn_search=1, **kw)
    Tb, Eb, ib = sc.icp_refine_ex(src, tgt, G0, nn_search=2, **kw)
    assert np.array_equal(Ta, Tb) and np.array_equal(Ea, Eb) and np.array_equal(ia, ib)
    assert (Ea < 1e-5).all()      # it did register (rms below 3 mm)
    # auto picks the grid at this size (20000 x 30000 tests per pose) and agrees as well
    Tc, Ec, ic = sc.icp_refine_ex(src, tgt, G0, nn_search=0, **kw)
    assert np.array_equal(Ta, Tc) and np.array_equal(ia, ic)


def test_table_icp_shape_target_beyond_16_bit_positions():
    """SceneCfg.cpp:101,135-141: the segmented scene (30 000 points) aligned to table.ply (100 000 points: beyond the
    16-bit positions of the LDS index), max correspondence distance 1 cm.  The capped search runs on the uniform grid
    (32-bit cell starts, any target size); transforms, energies and iteration counts equal the exhaustive scan's."""
    rng = np.random.default_rng(12)
    # a table top with a rim: 100 000 points on a 1.2 m x 0.8 m plane + edge strips
    top = np.c_[rng.uniform(-0.6, 0.6, 90000), rng.uniform(-0.4, 0.4, 90000), 0.0005 * rng.standard_normal(90000)]
    rim = np.c_[rng.uniform(-0.6, 0.6, 10000), np.where(rng.random(10000) < 0.5, -0.4, 0.4), rng.uniform(-0.05, 0.0, 10000)]
    tgt = np.concatenate([top, rim]).astype(np.float32)
    assert len(tgt) > 65535
    R = synth._random_rot(rng, np.deg2rad(1.0))
    pick = rng.choice(len(tgt), 30000, replace=False)
    src = (tgt[pick] @ R.T + np.array([0.004, -0.003, 0.002]) + 0.0008 * rng.standard_normal((30000, 3))).astype(np.float32)
    src[:300] += rng.uniform(-0.2, 0.2, (300, 3)).astype(np.float32)     # objects standing on the table: beyond the cap
    G0 = synth.colmajor16(np.eye(4))[None]
    sc = LcpScorer()
    kw = dict(max_iterations=8, max_corr_dist=0.01, energy_ratio=0.0, transformation_epsilon=1e-9, absolute_mse=1e-12)
    Ta, Ea, ia = sc.icp_refine_ex(src, tgt, G0, nn_search=1, **kw)       # exhaustive scan: 3e9 tests per iteration
    Tb, Eb, ib = sc.icp_refine_ex(src, tgt, G0, nn_search=0, **kw)       # default: the grid at this size
    assert np.array_equal(Ta, Tb) and np.array_equal(Ea, Eb) and np.array_equal(ia, ib)
    assert Ea[0] < 1e-5 and ia[0] >= 2      # registered: rms below 3 mm (the noise is 0.8 mm per axis)


@pytest.mark.parametrize("far", [False, True])
def test_uncapped_icp_on_a_target_beyond_16_bit_positions(far):
    """No correspondence cap and a 100 000-point target (beyond the exact index): the uniform grid settles every query
    with a neighbour within its safe radius and the exhaustive scan only the others (icp_nn_grid_open +
    icp_nn_split<true>, round 4).  Trimmed and plain forms, two poses, from close by (nearly every query settled by the
    grid) and from 8 cm off (nearly none at first): transforms, energies and iteration counts equal the scan's, bit for bit."""
    rng = np.random.default_rng(21)
    top = np.c_[rng.uniform(-0.6, 0.6, 90000), rng.uniform(-0.4, 0.4, 90000), 0.0005 * rng.standard_normal(90000)]
    rim = np.c_[rng.uniform(-0.6, 0.6, 10000), np.where(rng.random(10000) < 0.5, -0.4, 0.4), rng.uniform(-0.05, 0.0, 10000)]
    tgt = np.concatenate([top, rim]).astype(np.float32)
    R = synth._random_rot(rng, np.deg2rad(1.0))
    pick = rng.choice(len(tgt), 6000, replace=False)
    off = np.array([0.05, -0.04, 0.05]) if far else np.array([0.004, -0.003, 0.002])
    src = (tgt[pick] @ R.T + off + 0.0008 * rng.standard_normal((6000, 3))).astype(np.float32)
    src[:100] += rng.uniform(-0.3, 0.3, (100, 3)).astype(np.float32)     # clutter far from the table
    src[7] = np.nan                                                      # a non-finite point has no neighbour either way
    G0 = np.stack([synth.colmajor16(np.eye(4)), synth.colmajor16(synth._se3(synth._random_rot(rng, np.deg2rad(0.5)), [0.002, 0.0, -0.001]))])
    sc = LcpScorer()
    for kw in (dict(max_iterations=6, trim_fraction=0.9, energy_ratio=0.0, transformation_epsilon=1e-9, absolute_mse=1e-12),
               dict(max_iterations=4, energy_ratio=0.0, transformation_epsilon=1e-9, absolute_mse=1e-12)):
        Ta, Ea, ia = sc.icp_refine_ex(src, tgt, G0, nn_search=1, **kw)       # exhaustive scan
        Tb, Eb, ib = sc.icp_refine_ex(src, tgt, G0, nn_search=0, **kw)       # default: open grid + scan of the rest
        assert np.array_equal(Ta, Tb) and np.array_equal(Ea, Eb) and np.array_equal(ia, ib)
        assert np.isfinite(Ta).all() and (ia >= 1).all()


def test_uncapped_icp_big_target_point_to_plane_and_nothing_settled():
    """The open grid on a 65 536-point target (the first size beyond the exact index): point-to-plane with normals, eight
    poses; and a segment so far off that the grid settles NOTHING (every query goes to the listed scan) -- equal to the scan."""
    rng = np.random.default_rng(33)
    n = 65536
    tgt = np.c_[rng.uniform(-0.5, 0.5, n), rng.uniform(-0.5, 0.5, n), 0.02 * np.sin(6 * rng.uniform(-0.5, 0.5, n))].astype(np.float32)
    nrm = np.tile(np.array([0, 0, 1], np.float32), (n, 1))
    src = (tgt[rng.choice(n, 3000, replace=False)] + np.array([0.003, 0.002, 0.004]) + 0.0005 * rng.standard_normal((3000, 3))).astype(np.float32)
    G0 = np.stack([synth.colmajor16(synth._se3(synth._random_rot(rng, np.deg2rad(0.4)), 0.002 * rng.standard_normal(3))) for _ in range(8)])
    sc = LcpScorer()
    kw = dict(max_iterations=5, error_metric=1, energy_ratio=0.0, transformation_epsilon=1e-9, absolute_mse=1e-12)
    Ta, Ea, ia = sc.icp_refine_ex(src, tgt, G0, tgt_nrm=nrm, nn_search=1, **kw)
    Tb, Eb, ib = sc.icp_refine_ex(src, tgt, G0, tgt_nrm=nrm, nn_search=0, **kw)
    assert np.array_equal(Ta, Tb) and np.array_equal(Ea, Eb) and np.array_equal(ia, ib)
    far = (src + np.array([0.0, 0.0, 0.6])).astype(np.float32)           # 60 cm above the sheet: no cell of the grid is near
    kw = dict(max_iterations=3, trim_fraction=0.8, energy_ratio=0.0)
    Ta, Ea, ia = sc.icp_refine_ex(far, tgt, G0[:2], nn_search=1, **kw)
    Tb, Eb, ib = sc.icp_refine_ex(far, tgt, G0[:2], nn_search=0, **kw)
    assert np.array_equal(Ta, Tb) and np.array_equal(Ea, Eb) and np.array_equal(ia, ib)


def test_scene_sized_source_sums_by_block_equal_the_single_workgroup_walk(monkeypatch):
    """Scenes beyond 4096 points without trimming: the iteration's f64 sums are formed per block of 4096 points by a
    workgroup each (icp_sums_partial) and added in block order; PGP_ICP_PART=0 keeps the one-workgroup walk.  The two
    associate the sums differently beyond one block: same iteration counts, transforms within 1e-6, energies within 1e-6
    relative -- and a cloud of at most one block is bit-equal (it IS the same tree)."""
    rng = np.random.default_rng(5)
    tgt = np.c_[rng.uniform(-0.5, 0.5, 40000), rng.uniform(-0.4, 0.4, 40000), 0.01 * rng.standard_normal(40000)].astype(np.float32)
    sc = LcpScorer()
    for n_src, exact in ((20000, False), (3000, True)):
        R = synth._random_rot(rng, np.deg2rad(0.8))
        src = (tgt[rng.choice(len(tgt), n_src, replace=False)] @ R.T + np.array([0.003, -0.002, 0.002])).astype(np.float32)
        G0 = np.stack([synth.colmajor16(np.eye(4)), synth.colmajor16(synth._se3(synth._random_rot(rng, np.deg2rad(0.3)), [0.001, 0.0, 0.001]))])
        for kw in (dict(max_iterations=12, max_corr_dist=0.02, energy_ratio=0.0, transformation_epsilon=1e-10, absolute_mse=1e-14, nn_search=2),
                   dict(max_iterations=6, energy_ratio=0.0, nn_search=1)):
            monkeypatch.setenv("PGP_ICP_PERSIST", "0")        # the host-driven path for both sizes
            monkeypatch.delenv("PGP_ICP_PART", raising=False)
            Ta, Ea, ia = sc.icp_refine_ex(src, tgt, G0, **kw)
            monkeypatch.setenv("PGP_ICP_PART", "0")
            Tb, Eb, ib = sc.icp_refine_ex(src, tgt, G0, **kw)
            assert np.array_equal(ia, ib)
            if exact:
                assert np.array_equal(Ta, Tb) and np.array_equal(Ea, Eb)
            else:
                assert np.abs(Ta - Tb).max() < 1e-6 and np.allclose(Ea, Eb, rtol=1e-6, atol=1e-14)


def test_old_entry_point_is_the_trimmed_form():
    g = np.load(GOLD)
    sc = LcpScorer()
    G0 = np.stack([synth.colmajor16(G) for G in g["guesses"]])
    a = sc.icp_refine(g["seg_c"], g["model"], G0, trim=0.9, max_iterations=60)
    b = sc.icp_refine_ex(g["seg_c"], g["model"], G0, max_iterations=60, trim_fraction=0.9, energy_ratio=1.0)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_point_to_plane_needs_normals():
    from physimglobalpose_amd._lib import PgpError
    g = np.load(GOLD)
    sc = LcpScorer()
    with pytest.raises(PgpError):
        sc.icp_refine_ex(g["seg"], g["model"], synth.colmajor16(g["guesses"][0])[None], error_metric=1)


@pytest.mark.parametrize("metric", [0, 1])
def test_scene_sized_capped_icp_in_one_launch_equals_the_host_driven_iterations(metric, monkeypatch):
    """Round 5 (csrc/icp.hip icp_scene_persist): the scene-sized capped form -- the reference's one live ICP call,
    SceneCfg.cpp:101,135-141 -- runs every iteration inside ONE cooperative launch (chunk / unit / pose tickets).
    PGP_ICP_SCENE_PERSIST=0 keeps the host-driven iterations (icp_nn_grid + icp_sums_partial + icp_refine per
    iteration): transforms, energies and iteration counts must be theirs bit for bit -- several poses that stop at
    different iterations, a cloud that ends inside a unit and inside a block, point-to-point and point-to-plane."""
    rng = np.random.default_rng(31 + metric)
    n_tgt, n_src = 60000, 20011
    tgt = np.c_[rng.uniform(-0.5, 0.5, n_tgt), rng.uniform(-0.4, 0.4, n_tgt), 0.004 * np.sin(7 * rng.uniform(-1, 1, n_tgt))]
    tgt[:, 2] += 0.05 * tgt[:, 0] ** 2                                       # a gently curved sheet: the plane metric has something to hold
    tgt = tgt.astype(np.float32)
    nrm = np.c_[-0.1 * tgt[:, 0], np.zeros(n_tgt), np.ones(n_tgt)]
    nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
    R = synth._random_rot(rng, np.deg2rad(0.8))
    src = (tgt[rng.choice(n_tgt, n_src, replace=False)] @ R.T + np.array([0.003, -0.002, 0.002]) + 0.0005 * rng.standard_normal((n_src, 3))).astype(np.float32)
    src[:200] += rng.uniform(-0.2, 0.2, (200, 3)).astype(np.float32)          # beyond the cap
    src[11] = np.nan
    G0 = np.stack([synth.colmajor16(np.eye(4)),
                   synth.colmajor16(synth._se3(synth._random_rot(rng, np.deg2rad(0.4)), [0.002, 0.0, -0.001])),
                   synth.colmajor16(synth._se3(R.T, -R.T @ np.array([0.003, -0.002, 0.002]))),     # starts at the answer: stops first
                   synth.colmajor16(synth._se3(np.eye(3), [0.5, 0.5, 0.5]))])                       # nothing within the cap: no pairs
    sc = LcpScorer()
    for kw in (dict(max_iterations=25, max_corr_dist=0.01, energy_ratio=0.0, transformation_epsilon=1e-9, absolute_mse=1e-12),
               dict(max_iterations=3, max_corr_dist=0.02, energy_ratio=0.0)):
        kw = dict(kw, nn_search=2, error_metric=metric)
        monkeypatch.setenv("PGP_ICP_SCENE_PERSIST", "0")
        Ta, Ea, ia = sc.icp_refine_ex(src, tgt, G0, tgt_nrm=nrm if metric else None, **kw)
        monkeypatch.delenv("PGP_ICP_SCENE_PERSIST")
        Tb, Eb, ib = sc.icp_refine_ex(src, tgt, G0, tgt_nrm=nrm if metric else None, **kw)
        assert np.array_equal(ia, ib), (ia, ib)
        assert np.array_equal(Ta, Tb) and np.array_equal(Ea, Eb)
        assert len(set(ia.tolist())) >= 2 or kw["max_iterations"] == 3        # the poses stop at different iterations
    # one pose, the table's shape (the bench row)
    Tc, Ec, ic = sc.icp_refine_ex(src, tgt, G0[:1], max_iterations=30, max_corr_dist=0.01, energy_ratio=0.0, transformation_epsilon=1e-9,
                                  absolute_mse=1e-12, error_metric=metric, tgt_nrm=nrm if metric else None)
    assert ic[0] >= 1 and np.isfinite(Tc).all() and Ec[0] < 1e-4
