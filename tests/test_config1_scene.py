"""BASELINE.json configs[0]: "test-scene/ single object, 256 hypotheses, PCS+LCP on reference CPU
path (plumbing, no GPU)".  The segment clouds come from the reference's own test-scene/ frame
(tests/golden/test_scene_segments.npz, derived by tests/golden/make_golden.py: depth decode,
mask, back-projection, 1 cm voxel grid, normals).  The object models are not shipped with the
reference (README steps 2-3), so a stand-in model is posed around each segment.

CPU part: the oracle's kd-tree path and its exhaustive path agree on every hypothesis (the
plumbing the config names).  GPU part: the HIP path gives the same scores on the same data."""
import os

import numpy as np
import pytest

from physimglobalpose_amd import synth
from _checkers import Oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "test_scene_segments.npz")


def _problem(cls):
    g = np.load(GOLD)
    seg, nrm = g[f"seg_{cls}"], g[f"nrm_{cls}"]
    rng = np.random.default_rng(int(cls))
    Qw, Qn = synth.make_model(rng, 1000)
    Qw, Qn = Qw.astype(np.float32), Qn.astype(np.float32)
    # centre as Match4PCSBase::init does (stand-in search cloud = a 200-point subset)
    from physimglobalpose_amd import _lib
    import ctypes as C
    P, Qs, Qv = seg.copy(), Qw[:200].copy(), Qw.copy()
    cP, cQ = np.zeros(3, np.float32), np.zeros(3, np.float32)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    assert _lib.load().pgp_center(fp(P), len(P), fp(Qs), len(Qs), fp(Qv), len(Qv), fp(cP), fp(cQ)) == 0
    T = []
    for k in range(256):     # 256 hypotheses: random rotations, centroid-to-centroid +- 3 cm
        M = synth._se3(synth._random_rot(rng), 0.03 * rng.standard_normal(3) if k else np.zeros(3))
        T.append(synth.colmajor16(M))
    w = np.ones(len(P), np.float32)
    return P, nrm, w, Qv, Qn, np.stack(T)


@pytest.mark.parametrize("cls", [2, 3, 8])
def test_cpu_path_kd_and_exhaustive_agree(cls):
    P, Pn, w, Q, Qn, T = _problem(cls)
    kd = Oracle(P, Pn, w, Q, Qn, use_kd=True)
    bf = Oracle(P, Pn, w, Q, Qn, use_kd=False)
    for mode in (0, 1):
        a, bi_a, sel_a = kd.score_batch(T, 0.005, mode=mode)
        b, bi_b, sel_b = bf.score_batch(T, 0.005, mode=mode)
        assert np.array_equal(a, b) and bi_a == bi_b and np.array_equal(sel_a, sel_b)
    assert a.max() > 0                     # some stand-in pose overlaps the real segment


@pytest.mark.gpu
@pytest.mark.parametrize("cls", [2, 3, 8])
def test_hip_path_matches_on_the_real_segment(cls):
    from physimglobalpose_amd import LcpScorer, PGP_MODE_PLAIN, PGP_MODE_WEIGHTED
    P, Pn, w, Q, Qn, T = _problem(cls)
    orc = Oracle(P, Pn, w, Q, Qn)
    sc = LcpScorer()
    sc.init(P, Pn, w, Q, Qn, 0.005)
    s, c, bi, bs = sc.score(T, PGP_MODE_PLAIN)
    so, bio, _ = orc.score_batch(T, 0.005, mode=0)
    assert np.array_equal(s, so) and bi == bio
    s, c, bi, bs = sc.score(T, PGP_MODE_WEIGHTED)
    so, bio, _ = orc.score_batch(T, 0.005, mode=1)
    assert np.allclose(s, so, rtol=0, atol=2e-6)
