"""Exact nearest-neighbour index over the static ICP target (csrc/icp.hip: nnidx_*, icp_nn_index,
icp_persist_index) against the exhaustive scan (the checker path, PGP_ICP_NN=scan): transforms,
energies and iteration counts must be identical BIT FOR BIT -- the index only changes which target
points a query looks at, never the answer (smallest d2, then lowest original index) -- in every
ICP form of the call sites (UCTState.cpp:121-204, utilities.cpp:651-838, State.cpp:139-145)."""
import numpy as np
import pytest

from physimglobalpose_amd import LcpScorer, synth
from _checkers import oracle_icp

pytestmark = pytest.mark.gpu


def _problem(seed, n_model, n_seg, n_guess, rot_deg=8.0, trans=0.01, noise=0.0005, outliers=0.0):
    rng = np.random.default_rng(seed)
    M, N = synth.make_model(rng, n_model)
    M = M.astype(np.float32)
    R = synth._random_rot(rng)
    t = rng.uniform(-0.2, 0.2, 3) + np.array([0, 0, 0.8])
    seg = M[rng.choice(n_model, n_seg, replace=n_seg > n_model)]
    S = seg @ R.T + t + noise * rng.standard_normal(seg.shape)
    n_out = int(outliers * n_seg)
    if n_out:   # segmentation bleed: points up to 15 cm away from the object
        S[rng.choice(n_seg, n_out, replace=False)] += rng.uniform(-0.15, 0.15, (n_out, 3))
    S = S.astype(np.float32)
    Tinv = np.linalg.inv(synth._se3(R, t))
    G = np.stack([synth.colmajor16(Tinv @ synth._se3(synth._random_rot(rng, np.deg2rad(rot_deg)),
                                                     trans * rng.standard_normal(3)))
                  for _ in range(n_guess)])
    return S, M, N.astype(np.float32), G


def _three_ways(monkeypatch, S, M, G, nrm=None, **opts):
    """scan (checker), index with one persistent workgroup per pose, index on the host-driven path"""
    sc = LcpScorer()
    monkeypatch.setenv("PGP_ICP_NN", "scan")
    ref = sc.icp_refine_ex(S, M, G, tgt_nrm=nrm, **opts)
    monkeypatch.setenv("PGP_ICP_NN", "index")
    monkeypatch.setenv("PGP_ICP_PERSIST", "1")
    a = sc.icp_refine_ex(S, M, G, tgt_nrm=nrm, nn_search=3, **opts)
    monkeypatch.setenv("PGP_ICP_PERSIST", "0")
    b = sc.icp_refine_ex(S, M, G, tgt_nrm=nrm, nn_search=3, **opts)
    for name, got in (("persistent", a), ("split", b)):
        for x, y, what in zip(ref, got, ("T", "energy", "iters")):
            assert np.array_equal(x, y), (name, what, np.abs(np.asarray(x, np.float64) - y).max())
    return ref


FORMS = {
    "trimmed": dict(max_iterations=40, trim_fraction=0.9, energy_ratio=1.0),
    "trim_half": dict(max_iterations=25, trim_fraction=0.5, energy_ratio=1.0),
    "all_points": dict(max_iterations=25, trim_fraction=1.0, energy_ratio=1.0),
    "capped": dict(max_iterations=50, max_corr_dist=0.02, energy_ratio=0.0, transformation_epsilon=1e-8,
                   absolute_mse=1e-12),
    "plain_pcl": dict(max_iterations=30, energy_ratio=0.0, transformation_epsilon=0.0, absolute_mse=1e-12),
    "plane": dict(max_iterations=30, energy_ratio=0.0, error_metric=1, transformation_epsilon=0.0,
                  absolute_mse=1e-12),
    "pointmatcher": dict(max_iterations=40, trim_fraction=0.75, energy_ratio=0.0, transformation_epsilon=-1.0,
                         absolute_mse=-1.0, min_diff_rot=0.001, min_diff_trans=0.005, smooth_length=4),
}


@pytest.mark.parametrize("form", list(FORMS))
def test_index_equals_scan_in_every_form(form, monkeypatch):
    S, M, N, G = _problem(21, 3000, 1700, 6)
    T, e, it = _three_ways(monkeypatch, S, M, G, nrm=N, **FORMS[form])
    assert (it >= 1).all()


def test_with_far_outliers_and_model_sized_problem(monkeypatch):
    """configs[2] shape: 5000-point model, 2500-point segment of which 10 % are up to 15 cm off the object."""
    S, M, N, G = _problem(22, 5000, 2500, 8, rot_deg=5.0, trans=0.005, outliers=0.10)
    T, e, it = _three_ways(monkeypatch, S, M, G, **FORMS["trimmed"])
    monkeypatch.delenv("PGP_ICP_NN")
    monkeypatch.delenv("PGP_ICP_PERSIST")
    To, eo, ito = oracle_icp(S, M, G, trim=0.9, max_iterations=40)
    assert np.array_equal(it, ito) and np.abs(T - To).max() < 2e-6


def test_source_beyond_one_workgroups_registers_uses_the_split_index_path(monkeypatch):
    S, M, N, G = _problem(23, 4000, 6000, 3, rot_deg=4.0, trans=0.004, outliers=0.05)
    _three_ways(monkeypatch, S, M, G, **FORMS["trimmed"])


def test_ties_duplicates_and_nonfinite_points(monkeypatch):
    """Duplicate target points (exact distance ties: the lowest ORIGINAL index must win, whatever the
    cell order), source points that coincide with target points, a NaN and a huge source point."""
    rng = np.random.default_rng(24)
    M = rng.uniform(-0.1, 0.1, (1200, 3)).astype(np.float32)
    M[600:900] = M[:300]                          # every one of these exists twice
    S = M[rng.choice(1200, 500)].copy()
    S[7] = np.nan
    S[11] = 3e18
    S[13] = [5.0, -4.0, 2.0]                      # far outside the grid
    G = np.stack([synth.colmajor16(np.eye(4)),
                  synth.colmajor16(synth._se3(synth._random_rot(rng, np.deg2rad(2.0)), [0.002, 0, -0.001]))])
    _three_ways(monkeypatch, S, M, G, max_iterations=10, trim_fraction=0.8, energy_ratio=0.0)


def test_tiny_flat_and_degenerate_targets(monkeypatch):
    rng = np.random.default_rng(25)
    for M in (rng.uniform(-0.05, 0.05, (3, 3)),                                     # three points
              np.c_[rng.uniform(-0.1, 0.1, (400, 2)), np.zeros(400)],              # a plane z = 0
              np.repeat(rng.uniform(-0.1, 0.1, (1, 3)), 50, axis=0)):               # one point, 50 times
        M = M.astype(np.float32)
        S = (M[rng.choice(len(M), 64)] + 0.001 * rng.standard_normal((64, 3))).astype(np.float32)
        G = synth.colmajor16(synth._se3(synth._random_rot(rng, np.deg2rad(3.0)), [0.001, 0.002, 0]))[None]
        _three_ways(monkeypatch, S, M, G, max_iterations=8, trim_fraction=0.9, energy_ratio=0.0)


def test_default_path_is_the_index_and_matches_oracle():
    """No environment override, no option: pgp_icp_refine goes through the index."""
    S, M, N, G = _problem(26, 2500, 1200, 16, rot_deg=6.0, trans=0.008, noise=0.0003)
    sc = LcpScorer()
    T, e, it = sc.icp_refine(S, M, G, trim=0.9, max_iterations=60)
    To, eo, ito = oracle_icp(S, M, G, trim=0.9, max_iterations=60)
    assert np.array_equal(it, ito) and np.abs(T - To).max() < 2e-6 and np.allclose(e, eo, rtol=1e-4, atol=1e-12)


def test_targets_beyond_lds_use_the_same_index_from_l2(monkeypatch):
    """A 12 000-point target does not fit a CU's LDS beside the per-query arrays: the image stays in HBM / L2 and
    the same search reads it there -- still bit-identical to the scan; the 16-bit tables end at 65 535 points."""
    from physimglobalpose_amd._lib import PgpError
    rng = np.random.default_rng(27)
    M = synth.make_model(rng, 12000)[0].astype(np.float32)
    R = synth._random_rot(rng)
    S = (M[rng.choice(12000, 3000, replace=False)] @ R.T + [0.1, 0.0, 0.6] + 0.0005 * rng.standard_normal((3000, 3))).astype(np.float32)
    Tinv = np.linalg.inv(synth._se3(R, [0.1, 0.0, 0.6]))
    G = np.stack([synth.colmajor16(Tinv @ synth._se3(synth._random_rot(rng, np.deg2rad(4)), 0.004 * rng.standard_normal(3)))
                  for _ in range(5)])
    _three_ways(monkeypatch, S, M, G, max_iterations=12, trim_fraction=0.9, energy_ratio=1.0)
    # the LDS-sized problem through the L2 path as well (A/B knob)
    S2, M2, N2, G2 = _problem(31, 3000, 1700, 4)
    monkeypatch.setenv("PGP_ICP_IMAGE", "global")
    _three_ways(monkeypatch, S2, M2, G2, **FORMS["trimmed"])
    monkeypatch.delenv("PGP_ICP_IMAGE")
    sc = LcpScorer()
    big = rng.uniform(-0.2, 0.2, (70000, 3)).astype(np.float32)
    a = sc.icp_refine_ex(S[:200], big, G[:1], max_iterations=2, nn_search=0)      # auto: exhaustive scan
    b = sc.icp_refine_ex(S[:200], big, G[:1], max_iterations=2, nn_search=1)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    with pytest.raises(PgpError):
        sc.icp_refine_ex(S[:200], big, G[:1], max_iterations=2, nn_search=3)


def test_resident_index_is_reused_and_never_stale(monkeypatch):
    """Host-pointer calls hash the target: the same model keeps its index on the device across calls, a
    different model of the same size (or the same model after an in-place edit) gets a new one."""
    S, M, N, G = _problem(28, 2000, 900, 4)
    S2, M2, N2, G2 = _problem(29, 2000, 900, 4)
    sc = LcpScorer()
    kw = dict(max_iterations=15, trim_fraction=0.9, energy_ratio=1.0)
    monkeypatch.setenv("PGP_ICP_NN", "scan")
    ref1 = sc.icp_refine_ex(S, M, G, **kw)
    ref2 = sc.icp_refine_ex(S2, M2, G2, **kw)
    Me = M.copy()
    Me[::7] += np.float32(0.004)
    ref3 = sc.icp_refine_ex(S, Me, G, **kw)
    monkeypatch.setenv("PGP_ICP_NN", "index")
    for _ in range(2):   # second round: every call finds ANOTHER target's index resident
        for ref, args in ((ref1, (S, M, G)), (ref1, (S, M, G)), (ref2, (S2, M2, G2)), (ref3, (S, Me, G)), (ref1, (S, M, G))):
            got = sc.icp_refine_ex(*args, **kw)
            assert all(np.array_equal(x, y) for x, y in zip(ref, got))


@pytest.mark.parametrize("form", ["trimmed", "capped", "plane", "all_points"])
def test_several_workgroups_per_pose_give_the_same_bits(form, monkeypatch):
    """Few poses: 2 or 4 workgroups share a pose's search and meet once per iteration (icp_persist_index,
    cooperative launch).  Every workgroup then holds the same distances and correspondences, so transforms,
    energies and iteration counts equal the one-workgroup run bit for bit -- also when poses stop at different
    iterations, with 65 poses (two workgroups each) and with a source too short to share."""
    S, M, N, G = _problem(32, 3000, 1900, 9, rot_deg=7.0, trans=0.008, outliers=0.05)
    sc = LcpScorer()
    runs = {}
    for wgs in ("1", "2", "4"):
        monkeypatch.setenv("PGP_ICP_WGS", wgs)
        runs[wgs] = sc.icp_refine_ex(S, M, G, tgt_nrm=N, **FORMS[form])
    for wgs in ("2", "4"):
        for x, y, what in zip(runs["1"], runs[wgs], ("T", "energy", "iters")):
            assert np.array_equal(x, y), (wgs, what)
    assert len(set(runs["1"][2].tolist())) > 1 or form != "trimmed"      # the poses do not stop together
    monkeypatch.delenv("PGP_ICP_WGS")
    if form == "trimmed":
        G65 = np.concatenate([G] * 8)[:65]
        auto = sc.icp_refine_ex(S, M, G65, **FORMS[form])                  # 65 poses: two workgroups each by default
        monkeypatch.setenv("PGP_ICP_WGS", "1")
        one = sc.icp_refine_ex(S, M, G65, **FORMS[form])
        short = sc.icp_refine_ex(S[:100], M, G[:2], **FORMS[form])          # 100 points: one workgroup
        monkeypatch.setenv("PGP_ICP_WGS", "4")
        short4 = sc.icp_refine_ex(S[:100], M, G[:2], **FORMS[form])
        for x, y in zip(auto + short, one + short4):
            assert np.array_equal(x, y)
        assert np.array_equal(auto[0][:9], runs["1"][0])


def test_helping_launch_gives_the_bits_of_the_plain_launch(monkeypatch):
    """PGP_ICP_HELP=1 (csrc/icp.hip HelpPub; off by default -- measured slower): workgroups that are through with their
    pose take search passes of the poses still running; who searches for a query cannot change its answer."""
    S, M, N, G = _problem(31, 4000, 2200, 140, rot_deg=7.0, trans=0.01, outliers=0.08)
    sc = LcpScorer()
    monkeypatch.setenv("PGP_ICP_HELP", "0")
    ref = sc.icp_refine(S, M, G, trim=0.9, max_iterations=25)
    monkeypatch.setenv("PGP_ICP_HELP", "1")
    for _ in range(3):
        got = sc.icp_refine(S, M, G, trim=0.9, max_iterations=25)
        for x, y in zip(ref, got):
            assert np.array_equal(x, y)
    assert len(np.unique(ref[2])) > 3      # poses of very different length: there was something to help with
