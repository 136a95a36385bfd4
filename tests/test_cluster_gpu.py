"""pgp_cluster_poses / pgp_pose_error (csrc/cluster.hip) against the oracle and the committed
fixture of the Eigen harness.  Pose errors: bit-equal floats are expected; the stated tolerance
is 1 ulp of float because atan2 / asin are the device's double routines, not glibc's.  Cluster
decisions are compared exactly after checking that no fixture pair sits within that band of a
threshold."""
import os

import numpy as np
import pytest

from _checkers import oracle_greedy_cluster, oracle_pose_error

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "cluster.npz")


@pytest.fixture(scope="module")
def sc():
    from physimglobalpose_amd import LcpScorer
    return LcpScorer(0)


def ulp_close(a, b, ulps=1):
    return np.all(np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64)) <= ulps)


def pose_set(rng, n, n_modes=10, spread=1.0):
    from scipy.spatial.transform import Rotation as Rot
    modes = [(Rot.random(random_state=int(rng.integers(1 << 30))), rng.uniform(-0.2, 0.2, 3)) for _ in range(n_modes)]
    T = np.zeros((n, 16), np.float32)
    for i in range(n):
        if rng.random() < 0.8:
            R0, t0 = modes[int(rng.integers(n_modes))]
            R = Rot.from_rotvec(rng.normal(0, np.radians(6.0 * spread), 3)) * R0
            t = t0 + rng.normal(0, 0.008 * spread, 3)
        else:
            R, t = Rot.random(random_state=int(rng.integers(1 << 30))), rng.uniform(-0.3, 0.3, 3)
        M = np.eye(4)
        M[:3, :3], M[:3, 3] = R.as_matrix(), t
        T[i] = M.astype(np.float32).ravel(order="F")
    return T


@pytest.mark.parametrize("k", [0, 1, 2])
def test_fixture(sc, k):
    g = np.load(GOLD)
    T, s, sym = g[f"T_{k}"], g[f"scores_{k}"], g[f"sym_{k}"]
    rot, trans = sc.pose_error(T[g[f"pair_a_{k}"]], T[g[f"pair_b_{k}"]], sym)
    assert ulp_close(rot, g[f"rot_{k}"]) and np.array_equal(trans, g[f"trans_{k}"])
    print("pose errors bit-equal:", np.array_equal(rot, g[f"rot_{k}"]))
    rep, assign = sc.cluster_poses(T, s, float(s.max()), sym)
    assert np.array_equal(rep, g[f"rep_{k}"])
    _, assign_o = oracle_greedy_cluster(T, s, float(s.max()), sym)
    assert np.array_equal(assign, assign_o)


@pytest.mark.parametrize("n,sym,seed", [(1, (0, 0, 0), 0), (63, (0, 0, 0), 1), (64, (90, 0, 0), 2), (65, (0, 180, 0), 3),
                                        (129, (360, 360, 360), 4), (1000, (0, 0, 0), 5), (2500, (180, 90, 0), 6)])
def test_random_sets_match_oracle(sc, n, sym, seed):
    rng = np.random.default_rng(seed)
    T = pose_set(rng, n)
    # LCP-like scores: many exact ties (k / |Q|), so the stable tie rule is exercised
    s = (rng.integers(0, 200, n).astype(np.float32) / np.float32(200)).astype(np.float32)
    best = float(s.max())
    rep, assign = sc.cluster_poses(T, s, best, sym)
    rep_o, assign_o = oracle_greedy_cluster(T, s, best, sym)
    assert np.array_equal(rep, rep_o)
    assert np.array_equal(assign, assign_o)


def test_more_than_64_words_per_row(sc):
    """m > 4096 candidates: rows of the bit matrix span more than one 64-word chunk."""
    rng = np.random.default_rng(77)
    T = pose_set(rng, 6000, n_modes=40)
    s = rng.random(6000).astype(np.float32) + np.float32(0.5)
    rep, assign = sc.cluster_poses(T, s, float(s.max()), (0, 0, 0), accept_fraction=0.0)
    rep_o, assign_o = oracle_greedy_cluster(T, s, float(s.max()), (0, 0, 0), accept_fraction=0.0)
    assert np.array_equal(rep, rep_o) and np.array_equal(assign, assign_o)
    assert len(rep) > 64


def test_parameters_and_edges(sc):
    rng = np.random.default_rng(9)
    T = pose_set(rng, 400, spread=2.0)
    s = rng.random(400).astype(np.float32)
    for frac, rt, tt in ((0.0, 10.0, 0.02), (0.9, 10.0, 0.02), (0.3, 25.0, 0.05), (0.5, 0.0, 0.0), (0.5, 1e9, 1e9)):
        rep, assign = sc.cluster_poses(T, s, float(s.max()), (0, 0, 0), frac, rt, tt)
        rep_o, assign_o = oracle_greedy_cluster(T, s, float(s.max()), (0, 0, 0), frac, rt, tt)
        assert np.array_equal(rep, rep_o) and np.array_equal(assign, assign_o)
    rep, assign = sc.cluster_poses(T, s, float(s.max()), (0, 0, 0), 0.5, 1e9, 1e9)
    assert len(rep) == 1 and rep[0] == int(np.argmax(s))        # everything collapses onto the best
    rep, assign = sc.cluster_poses(T[:0], s[:0], 0.0)
    assert len(rep) == 0 and len(assign) == 0
    rep, assign = sc.cluster_poses(T, np.zeros(400, np.float32), 0.0)  # nothing passes 0 > 0
    assert len(rep) == 0 and np.all(assign == -1)
    s_nan = s.copy()
    s_nan[::7] = np.nan                                           # NaN never passes the `>` test
    rep, assign = sc.cluster_poses(T, s_nan, float(np.nanmax(s_nan)))
    rep_o, assign_o = oracle_greedy_cluster(T, s_nan, float(np.nanmax(s_nan)))
    assert np.array_equal(rep, rep_o) and np.array_equal(assign, assign_o)


def test_pose_error_random_pairs(sc):
    rng = np.random.default_rng(21)
    A, B = pose_set(rng, 20000, spread=3.0), pose_set(rng, 20000, spread=3.0)
    B[::2, :12] = A[::2, :12]                                     # identical rotations: trace = 3 branch
    B[1::4] = A[1::4]                                             # identical poses
    for sym in ((0, 0, 0), (90, 180, 360)):
        rot, trans = sc.pose_error(A, B, sym)
        ro, to = oracle_pose_error(A, B, sym)
        assert np.array_equal(trans, to)
        assert ulp_close(rot, ro)
        print(sym, "rot bit-equal fraction", float(np.mean(rot == ro)))


def test_scored_hypotheses_end_to_end(sc):
    """Cluster what the scoring kernel produced (C2-like small case): the best-scored hypothesis is
    always the first representative and the ground-truth neighbourhood collapses."""
    from physimglobalpose_amd import LcpScorer, PGP_MODE_WEIGHTED, synth
    w = synth.make_workload(8000, 1200, 512, config_id=3)
    s2 = LcpScorer(0)
    s2.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    scores, _, bi, bs = s2.score(w.T, PGP_MODE_WEIGHTED, w.gate_deg)
    rep, assign = s2.cluster_poses(w.T, scores, bs)
    rep_o, assign_o = oracle_greedy_cluster(w.T, scores, bs)
    assert np.array_equal(rep, rep_o) and np.array_equal(assign, assign_o)
    assert rep[0] == bi
    assert len(rep) < int((scores > 0.5 * bs).sum())


def test_round_by_round_pass_equals_the_bit_matrix_pass(monkeypatch):
    """csrc/cluster.hip has two forms of the greedy pass: one round per representative (few clusters: the usual outcome of
    the reference's pruning, HypothesisSelection.cpp:70-77) and the pair-bit matrix + tile walk (many clusters).  Same
    representatives in the same order and the same assignment, whatever the number of clusters -- including the hand-over
    from the first form to the second (more than 64 clusters, or eight representatives that take less than half)."""
    from physimglobalpose_amd import LcpScorer, PGP_MODE_WEIGHTED, synth
    w = synth.make_workload(20000, 2000, 6000, config_id=77)
    sc = LcpScorer()
    sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    s, _, _, bs = sc.score(w.T, PGP_MODE_WEIGHTED, w.gate_deg)
    seen = set()
    for frac, rot, trans in ((0.5, 10.0, 0.02), (0.2, 10.0, 0.02), (0.05, 4.0, 0.01), (0.0, 10.0, 0.02), (0.0, 1.0, 0.002)):
        monkeypatch.setenv("PGP_CLUSTER_ROUNDS", "0")
        rep_m, asg_m = sc.cluster_poses(w.T, s, bs, accept_fraction=frac, rot_thresh_deg=rot, trans_thresh=trans)
        monkeypatch.delenv("PGP_CLUSTER_ROUNDS")
        rep_r, asg_r = sc.cluster_poses(w.T, s, bs, accept_fraction=frac, rot_thresh_deg=rot, trans_thresh=trans)
        assert np.array_equal(rep_m, rep_r) and np.array_equal(asg_m, asg_r), (frac, len(rep_m), len(rep_r))
        seen.add(len(rep_m) > 64)
    assert seen == {True, False}          # both sides of the hand-over were exercised
