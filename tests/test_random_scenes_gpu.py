"""Randomised scenes against the brute-force oracle: extents from centimetres to tens of metres, inlier radii from
1 mm to 5 cm, scenes far from the origin, clouds of awkward sizes, both forms of the index -- the scoring path
(index build, look-up, candidate tests, partial sums, arg-max) must return the oracle's counts and plain scores bit
for bit, the weighted scores within the stated tolerance and the same best index (base.cc:1700-1791, 1885-1908)."""
import numpy as np
import pytest

from physimglobalpose_amd import LcpScorer, PGP_MODE_PLAIN, PGP_MODE_WEIGHTED, synth
from _checkers import Oracle

pytestmark = pytest.mark.gpu
W_TOL = 2e-6


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    extent = float(10.0 ** rng.uniform(-1.3, 1.3))                 # 5 cm .. 20 m
    delta = float(10.0 ** rng.uniform(-3.0, -1.3))                 # 1 mm .. 5 cm
    offset = rng.uniform(-1.0, 1.0, 3) * float(10.0 ** rng.uniform(-1, 2))   # up to 100 m from the origin
    n_p = int(rng.integers(1, 6000))
    n_q = int(rng.integers(1, 700))
    shape = rng.uniform(0.05, 1.0, 3)                               # flat / elongated boxes too
    P = (rng.uniform(-0.5, 0.5, (n_p, 3)) * extent * shape + offset).astype(np.float32)
    if seed % 3 == 0:                                               # a surface: many points per cell
        P[:, 2] = np.float32(offset[2]) + (0.2 * delta * rng.standard_normal(n_p)).astype(np.float32)
    Pn = synth._unit(rng.standard_normal((n_p, 3))).astype(np.float32)
    Pw = rng.uniform(0.1, 1.0, n_p).astype(np.float32)
    src = P[rng.integers(0, n_p, n_q)].astype(np.float64)
    Q = (src + rng.normal(0, 0.4 * delta, (n_q, 3))).astype(np.float32)
    Qn = synth._unit(rng.standard_normal((n_q, 3))).astype(np.float32)
    c = Q.mean(0).astype(np.float64)
    T = []
    for i in range(int(rng.integers(1, 40))):
        R = synth._random_rot(rng, np.deg2rad(rng.uniform(0, 3.0) if i % 3 else 0.0))
        t = rng.normal(0, delta * (0.5 if i % 2 else 3.0), 3)
        T.append(synth.colmajor16(synth._se3(np.eye(3), c) @ synth._se3(R, t) @ synth._se3(np.eye(3), -c)))
    return P, Pn, Pw, Q, Qn, np.stack(T), delta


@pytest.mark.parametrize("seed", range(24))
def test_random_scene_matches_brute_force(seed, monkeypatch):
    P, Pn, Pw, Q, Qn, T, delta = _case(seed)
    orc = Oracle(P, Pn, Pw, Q, Qn, use_kd=False)
    expect = {m: orc.score_batch(T, delta, mode=m) for m in (PGP_MODE_PLAIN, PGP_MODE_WEIGHTED)}
    forms = ("auto", "sparse") if seed % 2 else ("auto", "dense")
    for form in forms:
        if form != "auto":
            monkeypatch.setenv("PGP_INDEX", form)
        sc = LcpScorer()
        sc.init(P, Pn, Pw, Q, Qn, delta)
        for mode in (PGP_MODE_PLAIN, PGP_MODE_WEIGHTED):
            s, c, bi, bs = sc.score(T, mode)
            so, bio, _ = expect[mode]
            if mode == PGP_MODE_PLAIN:
                assert np.array_equal(s, so), (seed, form, np.abs(s - so).max())
                assert np.array_equal(c, np.round(so.astype(np.float64) * len(Q)).astype(np.int32))
            else:
                assert np.allclose(s, so, rtol=0, atol=W_TOL), (seed, form, np.abs(s - so).max())
            assert bi == bio, (seed, form, mode, bi, bio)
        if form != "auto":
            monkeypatch.delenv("PGP_INDEX")
