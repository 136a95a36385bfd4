"""pgp_set_exact_ties: exact distance ties go to the scene point the reference's kd-tree returns
(KdTree::doQueryRestrictedClosestIndex, kdtree.h:394-459: `sqdist <= cl_dist`, the last candidate visited wins).
Checked against the oracle's restatement of that tree (pinned on the reference's own code by test_oracle_vs_ref) and
against the golden fixture `duplicates`, which the reference harness produced on a cloud whose points all exist twice."""
import os

import numpy as np
import pytest

from physimglobalpose_amd import LcpScorer, PGP_MODE_PLAIN, PGP_MODE_WEIGHTED, synth
from _checkers import Oracle

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
I16 = synth.colmajor16(np.eye(4))


def _exact(P, Pn, Pw, Q, Qn, delta, form=None, monkeypatch=None):
    if form:
        monkeypatch.setenv("PGP_INDEX", form)
    sc = LcpScorer()
    sc.set_exact_ties(True)
    sc.init(P, Pn, Pw, Q, Qn, delta)
    if form:
        monkeypatch.delenv("PGP_INDEX")
    return sc


def test_golden_duplicates_in_full():
    """The fixture test_golden_gpu has to skip half of: with the tree's tie rule the registered points, the weighted
    scores, their counts and the best hypothesis of the `duplicates` case are the reference's."""
    g = np.load(os.path.join(GOLD, "duplicates.npz"))
    sc = _exact(g["P"], g["Pn"], g["Pw"], g["Q"], g["Qn"], float(g["delta"]))
    s, c, bi, bs = sc.score(g["T"], PGP_MODE_PLAIN)
    assert np.array_equal(c, g["counts"]) and np.array_equal(s, g["scores"]) and bi == int(g["best_plain"])
    for h, T in enumerate(g["T"]):
        hits = g["hits"][h]
        assert np.array_equal(sc.registered(T, PGP_MODE_PLAIN), hits[hits >= 0])
    s, c, bi, bs = sc.score(g["T"], PGP_MODE_WEIGHTED, 30.0)
    assert np.allclose(s, g["wscores"], rtol=0, atol=2e-6)
    assert np.array_equal(c, np.diff(g["reg_off"]).astype(np.int32))
    assert bi == int(g["best_weighted"]) and abs(bs - g["wscores"][bi]) <= 2e-6
    for h, T in enumerate(g["T"]):
        reg = g["reg_flat"][g["reg_off"][h]:g["reg_off"][h + 1]]
        assert np.array_equal(sc.registered(T, PGP_MODE_WEIGHTED, 30.0), reg)
    # and without the option the lowest-index rule is visibly different on this fixture
    plain = LcpScorer()
    plain.init(g["P"], g["Pn"], g["Pw"], g["Q"], g["Qn"], float(g["delta"]))
    differs = any(not np.array_equal(plain.registered(T, PGP_MODE_PLAIN), g["hits"][h][g["hits"][h] >= 0])
                  for h, T in enumerate(g["T"]))
    assert differs


def _lattice_case(seed, n_side=14):
    """Scene points on a lattice of pitch 2^-8 m (every coordinate and every midpoint exact in float): model points
    at lattice points, edge midpoints and face centres -- 1, 2 and 4 scene points at exactly the same distance --
    plus duplicated scene points with different normals and weights."""
    rng = np.random.default_rng(seed)
    pitch = 2.0 ** -8
    g = np.stack(np.meshgrid(*[np.arange(n_side)] * 3, indexing="ij"), -1).reshape(-1, 3)
    keep = rng.random(len(g)) < 0.7
    P = (g[keep] * pitch).astype(np.float32)
    dup = P[rng.integers(0, len(P), len(P) // 5)]
    P = np.concatenate([P, dup])[rng.permutation(len(P) + len(dup))]
    Pn = synth._unit(rng.standard_normal(P.shape)).astype(np.float32)
    Pw = rng.uniform(0.1, 1.0, len(P)).astype(np.float32)
    cells = rng.integers(1, n_side - 2, (600, 3)).astype(np.float64)
    off = rng.choice([0.0, 0.5], (600, 3))                  # lattice point / edge midpoint / face centre / body centre
    Q = ((cells + off) * pitch).astype(np.float32)
    Qn = synth._unit(rng.standard_normal(Q.shape)).astype(np.float32)
    delta = np.float32(pitch * 0.9)                          # reaches body centres' 8 corners (0.866 pitch)
    # the identity and lattice translations keep every distance tie exact; one generic pose for contrast
    T = [I16]
    for k in range(5):
        T.append(synth.colmajor16(synth._se3(np.eye(3), rng.integers(-2, 3, 3) * pitch)))
    T.append(synth.colmajor16(synth._se3(synth._random_rot(rng, 0.01), [0.0003, 0, 0])))
    return P, Pn, Pw, Q, Qn, float(delta), np.stack(T)


@pytest.mark.parametrize("form", ["dense", "sparse"])
def test_lattice_ties_follow_the_tree(form, monkeypatch):
    P, Pn, Pw, Q, Qn, delta, T = _lattice_case(7)
    orc = Oracle(P, Pn, Pw, Q, Qn, use_kd=True)
    sc = _exact(P, Pn, Pw, Q, Qn, delta, form, monkeypatch)
    assert sc.index_info()["sparse"] == (1 if form == "sparse" else 0)
    n_tied = 0
    for h in range(len(T)):
        so, reg = orc.weighted_verify(T[h], delta, 30.0)
        got = sc.registered(T[h], PGP_MODE_WEIGHTED, 30.0)
        assert np.array_equal(got, reg), h
        _, _, hits = orc.verify(T[h], delta)
        assert np.array_equal(sc.registered(T[h], PGP_MODE_PLAIN), hits[hits >= 0]), h
        lo = LcpScorer()
    s, c, bi, bs = sc.score(T, PGP_MODE_WEIGHTED, 30.0)
    so, bio, _ = orc.score_batch(T, delta, mode=1, gate_deg=30.0)
    assert np.allclose(s, so, rtol=0, atol=2e-6) and bi == bio
    # the lowest-index rule gives other registrations here (the test would be vacuous otherwise)
    base = LcpScorer()
    base.init(P, Pn, Pw, Q, Qn, delta)
    _, reg0 = orc.weighted_verify(T[0], delta, 30.0)
    assert not np.array_equal(base.registered(T[0], PGP_MODE_WEIGHTED, 30.0), reg0)
    sb, _, _, _ = base.score(T, PGP_MODE_WEIGHTED, 30.0)
    assert np.abs(sb - so).max() > 1e-4


def test_exact_records_and_device_scene_with_ties(monkeypatch):
    """The tie rule reaches the exact-records pass and a scene handed over on the device."""
    import torch
    P, Pn, Pw, Q, Qn, delta, T = _lattice_case(11, n_side=10)
    orc = Oracle(P, Pn, Pw, Q, Qn, use_kd=True)
    sc = LcpScorer()
    sc.set_exact_ties(True)
    sc.set_scene_device(torch.from_numpy(P).cuda(), len(P), torch.from_numpy(Pn).cuda(), torch.from_numpy(Pw).cuda(), delta)
    sc.set_model(Q, Qn)
    sc.set_exact_records(True)
    Tb = np.concatenate([T] * 8)
    s, c, bi, bs = sc.score(Tb, PGP_MODE_WEIGHTED, 30.0)
    so, bio, sel = orc.score_batch(Tb, delta, mode=1, gate_deg=30.0)
    assert bi == bio and np.array_equal(LcpScorer.running_best(s), sel)
    assert np.array_equal(s[sel], so[sel])                  # records carry the reference's own sums, bit for bit
    sc.set_exact_ties(False)                                # off again: the next scene has no tree
    sc.set_scene(P, Pn, Pw, delta)
    s2, _, _, _ = sc.score(T, PGP_MODE_WEIGHTED, 30.0)
    assert np.abs(s2 - so[:len(T)]).max() > 1e-4


def test_golden_lattice_ties_from_the_reference_tree():
    """tests/golden/lattice_ties.npz holds what the REFERENCE's kd-tree (oracle/_ref, built from the reference's own
    kdtree.h) returned on a 2278-point lattice scene whose queries have 2-, 4- and 8-fold exact ties across leaves:
    the per-point NN ids, the registered lists and the weighted scores must be reproduced."""
    g = np.load(os.path.join(GOLD, "lattice_ties.npz"))
    sc = _exact(g["P"], g["Pn"], g["Pw"], g["Q"], g["Qn"], float(g["delta"]))
    s, c, bi, bs = sc.score(g["T"], PGP_MODE_PLAIN)
    assert np.array_equal(c, g["counts"]) and np.array_equal(s, g["scores"])
    s, c, bi, bs = sc.score(g["T"], PGP_MODE_WEIGHTED, 30.0)
    assert np.allclose(s, g["wscores"], rtol=0, atol=2e-6)
    assert np.array_equal(c, np.diff(g["reg_off"]).astype(np.int32))
    n_diff = 0
    base = LcpScorer()
    base.init(g["P"], g["Pn"], g["Pw"], g["Q"], g["Qn"], float(g["delta"]))
    for h, T in enumerate(g["T"]):
        hits = g["hits"][h]
        assert np.array_equal(sc.registered(T, PGP_MODE_PLAIN), hits[hits >= 0]), h
        reg = g["reg_flat"][g["reg_off"][h]:g["reg_off"][h + 1]]
        assert np.array_equal(sc.registered(T, PGP_MODE_WEIGHTED, 30.0), reg), h
        n_diff += int(not np.array_equal(base.registered(T, PGP_MODE_PLAIN), hits[hits >= 0]))
    assert n_diff >= 5          # the lowest-index rule answers differently on most of these transforms
