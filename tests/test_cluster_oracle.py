"""Hypothesis clustering: the C restatement (oracle/pgp_oracle.c: orc_pose_error,
orc_greedy_cluster) against the committed fixture produced by the Eigen harness, and -- in the
build container -- against the harness itself (utilities.cpp:514-548, HypothesisSelection.cpp:66-115)."""
import os

import numpy as np
import pytest

from _checkers import (have_ref, oracle_greedy_cluster, oracle_pose_error, ref_greedy_cluster, ref_pose_error)

GOLD = os.path.join(os.path.dirname(__file__), "golden", "cluster.npz")


def cases():
    g = np.load(GOLD)
    for k in range(3):
        yield k, {n: g[f"{n}_{k}"] for n in ("T", "scores", "sym", "rep", "pair_a", "pair_b", "rot", "trans")}


@pytest.mark.parametrize("k,c", list(cases()))
def test_pose_error_matches_fixture_bitwise(k, c):
    rot, trans = oracle_pose_error(c["T"][c["pair_a"]], c["T"][c["pair_b"]], c["sym"])
    assert np.array_equal(rot, c["rot"])
    assert np.array_equal(trans, c["trans"])


@pytest.mark.parametrize("k,c", list(cases()))
def test_greedy_cluster_matches_fixture(k, c):
    rep, assign = oracle_greedy_cluster(c["T"], c["scores"], float(c["scores"].max()), c["sym"])
    assert np.array_equal(rep, c["rep"])
    # structural properties of the greedy pass
    s = c["scores"]
    bar = np.float32(0.5) * s.max()
    assert np.all(np.diff(s[rep]) < 0)                       # clusteredHypothesisSet is score-sorted
    assert np.array_equal(assign >= 0, s > bar)               # exactly the pruned ones are unassigned
    assert np.array_equal(assign[rep], rep)                   # a representative absorbs itself
    assert set(np.unique(assign[assign >= 0])) == set(rep)
    members = np.where((assign >= 0) & (assign != np.arange(len(s))))[0]
    assert np.all(s[assign[members]] > s[members])            # absorbed by a better-scored pose
    rot, trans = oracle_pose_error(c["T"][members], c["T"][assign[members]], c["sym"])
    assert np.all((rot < 10) & (trans < 0.02))
    # no representative is within the thresholds of an earlier one
    for i in range(1, min(len(rep), 40)):
        rot, trans = oracle_pose_error(np.repeat(c["T"][rep[i]][None], i, 0), c["T"][rep[:i]], c["sym"])
        assert not np.any((rot < 10) & (trans < 0.02))


def test_ties_keep_index_order_and_edges():
    rng = np.random.default_rng(3)
    T = np.tile(np.eye(4, dtype=np.float32).ravel(order="F"), (6, 1))
    T[:, 12] = [0.0, 0.5, 0.001, 0.501, 1.0, 0.0]            # x translations: {0,2,5} {1,3} {4}
    s = np.array([0.4, 0.4, 0.4, 0.4, 0.1, 0.4], np.float32)  # 4 is pruned (0.1 <= 0.2)
    rep, assign = oracle_greedy_cluster(T, s, 0.4)
    assert rep.tolist() == [0, 1]
    assert assign.tolist() == [0, 1, 0, 1, -1, 0]
    rep, assign = oracle_greedy_cluster(T[:0], s[:0], 0.0)
    assert len(rep) == 0 and len(assign) == 0
    rep, assign = oracle_greedy_cluster(T, np.zeros(6, np.float32), 0.0)   # 0 > 0 is false: all pruned
    assert len(rep) == 0 and np.all(assign == -1)
    del rng


@pytest.mark.skipif(not have_ref(), reason="needs oracle/_ref (build container)")
def test_oracle_vs_harness_random_pairs():
    from scipy.spatial.transform import Rotation as Rot
    rng = np.random.default_rng(11)
    n = 3000
    A = np.tile(np.eye(4), (n, 1, 1))
    B = np.tile(np.eye(4), (n, 1, 1))
    A[:, :3, :3] = Rot.random(n, random_state=1).as_matrix()
    near = Rot.from_rotvec(rng.normal(0, 0.2, (n, 3))) * Rot.from_matrix(A[:, :3, :3])
    far = Rot.random(n, random_state=2)
    B[:, :3, :3] = np.where((np.arange(n) % 2 == 0)[:, None, None], near.as_matrix(), far.as_matrix())
    A[:, :3, 3] = rng.normal(0, 0.1, (n, 3))
    B[:, :3, 3] = A[:, :3, 3] + rng.normal(0, 0.015, (n, 3))
    a = A.astype(np.float32).transpose(0, 2, 1).reshape(n, 16)
    b = B.astype(np.float32).transpose(0, 2, 1).reshape(n, 16)
    for sym in ((0, 0, 0), (90, 180, 360), (360, 90, 180)):
        ro, to = oracle_pose_error(a, b, sym)
        rr, tr = ref_pose_error(a, b, sym)
        assert np.array_equal(ro, rr) and np.array_equal(to, tr)


@pytest.mark.skipif(not have_ref(), reason="needs oracle/_ref (build container)")
def test_oracle_vs_harness_clusters():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden import cluster_pose_set
    rng = np.random.default_rng(5)
    for n, sym in ((300, (0, 0, 0)), (450, (180, 180, 0))):
        T, s = cluster_pose_set(rng, n, n_modes=6)
        rep, _ = oracle_greedy_cluster(T, s, float(s.max()), sym)
        assert np.array_equal(rep, ref_greedy_cluster(T, s, float(s.max()), sym))
