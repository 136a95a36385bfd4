"""The CPU oracle against the committed golden vectors (tests/golden/*.npz, produced by
tests/golden/make_golden.py from the reference's own kd-tree + Eigen, oracle/ref_harness.cc).
Bit-exact: counts, per-point NN ids (incl. the kd-tree's tie behaviour), weighted scores,
registered ids, best index, running-best subsequence, early-out scores."""
import glob
import os

import numpy as np
import pytest

from _checkers import Oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "*.npz"))
               if os.path.basename(p).startswith(("scene_", "boundary", "normal_gate", "duplicates", "near_ties")))


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def test_fixtures_present():
    assert set(CASES) >= {"scene_0", "scene_1", "scene_2", "boundary", "normal_gate", "duplicates"}


@pytest.mark.parametrize("name", CASES)
def test_kd_oracle_matches_golden(name):
    g = load(name)
    orc = Oracle(g["P"], g["Pn"], g["Pw"], g["Q"], g["Qn"], use_kd=True)
    delta = float(g["delta"])
    for h, T in enumerate(g["T"]):
        s, cnt, hits = orc.verify(T, delta)
        assert cnt == g["counts"][h] and np.float32(s) == g["scores"][h]
        assert np.array_equal(hits, g["hits"][h])
        ws, reg = orc.weighted_verify(T, delta)
        assert np.float32(ws) == g["wscores"][h]
        assert np.array_equal(reg, g["reg_flat"][g["reg_off"][h]:g["reg_off"][h + 1]])
    for mode, key in ((0, "plain"), (1, "weighted")):
        sc, bi, sel = orc.score_batch(g["T"], delta, mode=mode, threads=2)
        assert np.array_equal(sc, g["scores"] if mode == 0 else g["wscores"])
        assert bi == int(g["best_" + key]) and np.array_equal(sel, g["sel_" + key])
    sc, _, _ = orc.score_batch(g["T"], delta, mode=0, early_out=True)
    assert np.array_equal(sc, g["early_out_scores"])


@pytest.mark.parametrize("name", CASES)
def test_brute_oracle_matches_golden(name):
    """The exhaustive-scan definition gives the same inlier SET as the kd-tree walk; the NN id can
    differ only on exact distance ties (fixture `duplicates` is built to have them)."""
    g = load(name)
    orc = Oracle(g["P"], g["Pn"], g["Pw"], g["Q"], g["Qn"], use_kd=False)
    delta = float(g["delta"])
    for h, T in enumerate(g["T"]):
        s, cnt, hits = orc.verify(T, delta)
        assert cnt == g["counts"][h]
        assert np.array_equal(hits >= 0, g["hits"][h] >= 0)
        if name != "duplicates":
            assert np.array_equal(hits, g["hits"][h])
            ws, reg = orc.weighted_verify(T, delta)
            assert np.float32(ws) == g["wscores"][h]


def test_boundary_fixture_exercises_both_sides():
    g = load("boundary")
    c = g["counts"]
    assert 0 < c.min() and c.max() < len(g["Q"])      # some points in, some out
    # identity transform: the exactly-at-delta pairs are inliers (d2 <= delta^2 inclusive)
    P, Q = g["P"].astype(np.float32), g["Q"].astype(np.float32)
    d = (Q[64::2] - P[64:96:2]).astype(np.float32)
    d2 = (d[:, 0] * d[:, 0] + (d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2])).astype(np.float32)
    eps2 = np.float32(g["delta"]) * np.float32(g["delta"])
    at = d2 == eps2
    assert at.any()
    assert (g["hits"][0][64::2][at] >= 0).all()


def test_normal_gate_fixture_has_nan_rejections():
    g = load("normal_gate")
    # identity: every model point has its twin at distance 0, yet some are rejected by the gate
    assert g["counts"][0] == len(g["Q"])
    n_reg = g["reg_off"][1] - g["reg_off"][0]
    assert 0 < n_reg < len(g["Q"])


def test_rigid_fit_oracle_matches_golden():
    """ComputeRigidTransformFromCongruentPair restatement vs the Eigen-backed harness: status,
    centred 4x4 and rms bit-exact; de-centred pose to 1e-6 (SVD polar factors in the reference)."""
    from _checkers import oracle_rigid_from_pairs
    g = np.load(os.path.join(GOLD, "rigid_fit.npz"))
    n = len(g["p"])
    ids = np.arange(4 * n).reshape(n, 4)
    T, pose, status, rms = oracle_rigid_from_pairs(g["p"].reshape(-1, 3), g["q"].reshape(-1, 3), ids, ids,
                                                   g["centroid_P"], g["centroid_Q"])
    assert np.array_equal(status, g["status"])
    ok = status == 1
    assert np.array_equal(T[ok], g["T"][ok]) and np.array_equal(rms[ok], g["rms"][ok])
    assert np.abs(pose[ok] - g["pose"][ok]).max() < 1e-6


def test_kd_oracle_matches_the_lattice_tie_fixture():
    """tests/golden/lattice_ties.npz (make_ties_golden.py): 2-, 4- and 8-fold exact ties across the leaves of a
    2278-point lattice scene, answered by the reference's own tree.  The oracle's restatement of the tree (build,
    partition, descent, kdtree.h:394-459,522-641) must return the same ids, registered lists and sums."""
    g = load("lattice_ties")
    orc = Oracle(g["P"], g["Pn"], g["Pw"], g["Q"], g["Qn"], use_kd=True)
    delta = float(g["delta"])
    n_tie_sensitive = 0
    brute = Oracle(g["P"], g["Pn"], g["Pw"], g["Q"], g["Qn"], use_kd=False)
    for h, T in enumerate(g["T"]):
        s, cnt, hits = orc.verify(T, delta)
        assert cnt == g["counts"][h] and np.float32(s) == g["scores"][h]
        assert np.array_equal(hits, g["hits"][h])
        ws, reg = orc.weighted_verify(T, delta)
        assert np.float32(ws) == g["wscores"][h]
        assert np.array_equal(reg, g["reg_flat"][g["reg_off"][h]:g["reg_off"][h + 1]])
        _, _, hb = brute.verify(T, delta)
        n_tie_sensitive += int(not np.array_equal(hb, hits))
    assert n_tie_sensitive >= 5     # the fixture does exercise the tie rule
