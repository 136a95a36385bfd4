"""Edge cases and size-independent properties of the HIP scoring path (through the C ABI)."""
import numpy as np
import pytest

from physimglobalpose_amd import LcpScorer, PGP_MODE_PLAIN, PGP_MODE_WEIGHTED, synth, _lib
from _checkers import Oracle

pytestmark = pytest.mark.gpu
I16 = synth.colmajor16(np.eye(4))


def _cloud(rng, n, lo=-0.2, hi=0.2):
    p = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    nn = synth._unit(rng.standard_normal((n, 3))).astype(np.float32)
    return p, nn


def test_empty_hypothesis_list_and_empty_clouds():
    rng = np.random.default_rng(0)
    P, Pn = _cloud(rng, 100)
    Q, Qn = _cloud(rng, 10)
    sc = LcpScorer()
    sc.init(P, Pn, None, Q, Qn, 0.005)
    s, c, bi, bs = sc.score(np.zeros((0, 16), np.float32))
    assert len(s) == 0 and bi == -1 and bs == 0.0        # "returning identity" case, base.cc:1791
    sc2 = LcpScorer()                                     # empty scene: nothing can be an inlier
    sc2.init(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), None, Q, Qn, 0.005)
    s, c, bi, bs = sc2.score(np.stack([I16] * 3))
    assert not s.any() and not c.any() and bi == -1
    s, c, bi, bs = sc2.score(np.stack([I16] * 3), PGP_MODE_WEIGHTED)
    assert not s.any() and bi == -1


def test_state_errors_are_loud():
    rng = np.random.default_rng(1)
    P, Pn = _cloud(rng, 50)
    sc = LcpScorer()
    with pytest.raises(_lib.PgpError):
        sc.score(np.stack([I16]))                         # nothing set
    sc.set_scene(P, None, None, 0.005)
    sc.set_model(P[:5], None)
    sc.score(np.stack([I16]))                             # plain works without normals
    with pytest.raises(_lib.PgpError):
        sc.score(np.stack([I16]), PGP_MODE_WEIGHTED)      # weighted needs normals
    with pytest.raises(_lib.PgpError):
        sc.set_scene(P, None, None, -1.0)
    with pytest.raises(_lib.PgpError):
        sc.set_scene(P, None, None, float("nan"))


def test_identity_on_subset_gives_one():
    """Q_val subset of P under the identity: plain LCP = 1.0 exactly (SURVEY 8c sanity value)."""
    rng = np.random.default_rng(2)
    P, Pn = _cloud(rng, 3000)
    sc = LcpScorer()
    sc.init(P, Pn, None, P[::3], Pn[::3], 0.005)
    s, c, bi, bs = sc.score(np.stack([I16]))
    assert s[0] == 1.0 and c[0] == 1000 and bi == 0


def test_nonfinite_and_far_transforms_score_zero():
    rng = np.random.default_rng(3)
    P, Pn = _cloud(rng, 2000)
    Q, Qn = _cloud(rng, 300, -0.05, 0.05)
    T = np.stack([I16] * 6).copy()
    T[1, 12] = np.nan
    T[2, 13] = np.inf
    T[3, 14] = 1e30
    T[4, :] = np.nan
    T[5, 12] = -3e38
    sc = LcpScorer()
    sc.init(P, Pn, None, Q, Qn, 0.005)
    orc = Oracle(P, Pn, np.ones(len(P), np.float32), Q, Qn)
    for mode in (PGP_MODE_PLAIN, PGP_MODE_WEIGHTED):
        s, c, bi, bs = sc.score(T, mode)
        so, bio, _ = orc.score_batch(T, 0.005, mode=mode)
        assert np.array_equal(s[1:], np.zeros(5, np.float32)) and np.array_equal(so[1:], s[1:])
        assert bi == bio


@pytest.mark.parametrize("form", ["auto", "dense"])
def test_large_coordinates_and_large_extent(form, monkeypatch):
    """Scene far from the origin and wider than 1024 cells per axis.  By default the index takes its sparse
    form and keeps the 0.85 delta cell; held to the dense block array (PGP_INDEX=dense) the cell grows above
    delta.  Either way the dilated lists must still contain every inlier -> counts equal the oracle's."""
    rng = np.random.default_rng(4)
    P = rng.uniform(0, 12.0, (20000, 3)).astype(np.float32) + np.float32(40.0)
    Q = P[rng.integers(0, len(P), 500)] + rng.normal(0, 0.002, (500, 3)).astype(np.float32)
    Pn = synth._unit(rng.standard_normal(P.shape)).astype(np.float32)
    Qn = synth._unit(rng.standard_normal(Q.shape)).astype(np.float32)
    T = np.stack([I16] + [synth.colmajor16(synth._se3(synth._random_rot(rng, 0.002),
                                                      0.003 * rng.standard_normal(3))) for _ in range(7)])
    if form == "dense":
        monkeypatch.setenv("PGP_INDEX", "dense")
    sc = LcpScorer()
    sc.init(P, Pn, None, Q, Qn, 0.005)
    info = sc.index_info()
    if form == "dense":
        assert info["sparse"] == 0 and info["cell_size"] > 0.0055
        assert max(info["grid_nx"], info["grid_ny"], info["grid_nz"]) <= 1024
    else:
        assert info["sparse"] == 1 and abs(info["cell_size"] - 0.00425) < 1e-6 and info["grid_nx"] > 2800
    orc = Oracle(P, Pn, np.ones(len(P), np.float32), Q, Qn)
    s, c, bi, _ = sc.score(T)
    so, bio, _ = orc.score_batch(T, 0.005, mode=0)
    assert np.array_equal(s, so) and bi == bio and c.max() > 100


def test_scene_beyond_the_lattice_range_of_the_mantissa_trick():
    """The scoring kernel finds a cell as round(x * inv_h) - k0 in the mantissa of one fused multiply-add,
    which holds lattice numbers below 2^22.  A scene at 10^5 m with delta = 2 cm is at lattice number
    ~5.9 million: the library must notice (GridDesc.magic_ok) and score with the truncation kernel --
    same counts, scores and best index as the oracle, in both modes."""
    rng = np.random.default_rng(14)
    P = (rng.uniform(0, 3.0, (6000, 3)) + 1.0e5).astype(np.float32)
    Q = (P[rng.integers(0, len(P), 400)].astype(np.float64) + rng.normal(0, 0.006, (400, 3))).astype(np.float32)
    Pn = synth._unit(rng.standard_normal(P.shape)).astype(np.float32)
    Qn = synth._unit(rng.standard_normal(Q.shape)).astype(np.float32)
    w = rng.uniform(0, 1, len(P)).astype(np.float32)
    T = np.stack([I16] + [synth.colmajor16(synth._se3(np.eye(3), 0.01 * rng.standard_normal(3))) for _ in range(15)])
    sc = LcpScorer()
    sc.init(P, Pn, w, Q, Qn, 0.02)
    orc = Oracle(P, Pn, w, Q, Qn)
    s, c, bi, _ = sc.score(T)
    so, bio, _ = orc.score_batch(T, 0.02, mode=0)
    assert np.array_equal(s, so) and bi == bio and c.max() > 50
    s, c, bi, _ = sc.score(T, PGP_MODE_WEIGHTED, 30.0)
    so, bio, _ = orc.score_batch(T, 0.02, mode=1, gate_deg=30.0)
    assert np.allclose(s, so, rtol=0, atol=2e-6) and bi == bio


@pytest.mark.parametrize("delta", [0.001, 0.02, 0.1])
def test_other_radii(delta):
    w = synth.make_workload(4000, 400, 40, config_id=41)
    sc = LcpScorer()
    sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, delta)
    orc = Oracle(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm)
    s, c, bi, _ = sc.score(w.T)
    so, bio, _ = orc.score_batch(w.T, delta, mode=0)
    assert np.array_equal(s, so) and bi == bio
    s, c, bi, _ = sc.score(w.T, PGP_MODE_WEIGHTED)
    so, bio, _ = orc.score_batch(w.T, delta, mode=1)
    assert np.allclose(s, so, rtol=0, atol=2e-6)


def test_gate_angles():
    w = synth.make_workload(4000, 400, 24, config_id=42)
    sc = LcpScorer()
    sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    orc = Oracle(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm)
    for gate in (0.0, 5.0, 30.0, 89.9, 90.0, 91.0, 200.0):
        s, c, _, _ = sc.score(w.T, PGP_MODE_WEIGHTED, gate)
        so, _, _ = orc.score_batch(w.T, w.delta, mode=1, gate_deg=gate)
        assert np.allclose(s, so, rtol=0, atol=2e-6), gate
        for h in (0, 5):
            _, reg = orc.weighted_verify(w.T[h], w.delta, gate)
            assert np.array_equal(sc.registered(w.T[h], PGP_MODE_WEIGHTED, gate), reg)


def test_scene_and_model_can_be_replaced():
    a = synth.make_workload(3000, 300, 16, config_id=43)
    b = synth.make_workload(6000, 700, 16, config_id=44)
    sc = LcpScorer()
    for w in (a, b, a):
        sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
        so, bio, _ = Oracle(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm).score_batch(w.T, w.delta, mode=0)
        s, _, bi, _ = sc.score(w.T)
        assert np.array_equal(s, so) and bi == bio


def test_full_size_properties_c2():
    """BASELINE.json configs[1] at full size (50 000 x 5 000 x 4 096): properties that need no
    oracle pass over the whole batch, plus the oracle on every hypothesis."""
    w = synth.make_workload(50000, 5000, 4096, config_id=2)
    sc = LcpScorer()
    sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    s, c, bi, bs = sc.score(w.T)
    nQ = len(w.Q_xyz)
    assert c.min() >= 0 and c.max() <= nQ
    assert np.array_equal(s, (c.astype(np.float32) / np.float32(nQ)))
    assert bi == int(np.argmax(s)) and bs == s.max() and bi != 0
    # permuting the hypotheses permutes the scores; permuting the model changes nothing
    perm = np.random.default_rng(0).permutation(w.n_h)
    s2, c2, bi2, _ = sc.score(w.T[perm])
    assert np.array_equal(c2, c[perm]) and perm[bi2] == bi or s[perm[bi2]] == s[bi]
    qperm = np.random.default_rng(1).permutation(nQ)
    sc.set_model(w.Q_xyz[qperm], w.Q_nrm[qperm])
    s3, c3, bi3, _ = sc.score(w.T)
    assert np.array_equal(c3, c) and bi3 == bi
    # a sub-batch scores the same as inside the batch; repeated runs are bit-identical
    s4, c4, _, _ = sc.score(w.T[1000:1100])
    assert np.array_equal(c4, c[1000:1100])
    sw1 = sc.score(w.T, PGP_MODE_WEIGHTED)[0]
    sw2 = sc.score(w.T, PGP_MODE_WEIGHTED)[0]
    assert np.array_equal(sw1, sw2)
    assert (sw1 <= s + 1e-6).all()           # weights <= 1 and the gate only removes inliers
    # the oracle on EVERY hypothesis of the batch (the eight batches bench.py rotates through:
    # tests/test_bench_workload_parity_gpu.py)
    sc.set_model(w.Q_xyz, w.Q_nrm)
    orc = Oracle(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm)
    so, bio, _ = orc.score_batch(w.T, w.delta, mode=0, threads=8)
    assert np.array_equal(s, so) and bi == bio
    swo, _, _ = orc.score_batch(w.T, w.delta, mode=1, threads=8)
    assert np.allclose(sc.score(w.T, PGP_MODE_WEIGHTED)[0], swo, rtol=0, atol=2e-6)


def test_device_pointer_entry_matches_host_entry():
    import torch
    w = synth.make_workload(8000, 900, 300, config_id=45)
    sc = LcpScorer(0)
    sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    sc.reserve(300)
    dT = torch.from_numpy(w.T).cuda()
    ds = torch.zeros(300, dtype=torch.float32, device="cuda")
    dc = torch.zeros(300, dtype=torch.int32, device="cuda")
    db = torch.zeros(2, dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        sc.score_device(dT, ds, dc, db, mode=PGP_MODE_PLAIN, stream=side)
    side.synchronize()
    s, c, bi, bs = sc.score(w.T)
    assert np.array_equal(ds.cpu().numpy(), s) and np.array_equal(dc.cpu().numpy(), c)
    assert int(db[0]) == bi
    assert np.array([int(db[1])], np.int32).view(np.float32)[0] == np.float32(bs)
    with pytest.raises(_lib.PgpError):
        big = torch.zeros(301, 16, device="cuda")
        sc.score_device(big, torch.zeros(301, device="cuda"))   # beyond pgp_reserve: no hidden alloc


def test_large_and_odd_batch_sizes():
    """20 000 hypotheses (79 finalize workgroups folding one device-scope arg-max) down to 1, at
    sizes on both sides of the workgroup multiples: scores, counts and best against the oracle."""
    w = synth.make_workload(3000, 400, 20000, config_id=47)
    sc = LcpScorer()
    sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    orc = Oracle(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm)
    so, bio, _ = orc.score_batch(w.T, w.delta, mode=0, threads=8)
    for n in (20000, 16385, 16384, 1025, 1024, 1):
        s, c, bi, bs = sc.score(w.T[:n])
        assert np.array_equal(s, so[:n])
        exp = int(np.argmax(so[:n])) if so[:n].max() > 0 else -1
        assert bi == exp and (bi < 0 or np.float32(bs) == so[bi])
    swo, bwo, _ = orc.score_batch(w.T, w.delta, mode=1, gate_deg=w.gate_deg, threads=8)
    for n in (20000, 16384):
        sw, _, bi, _ = sc.score(w.T[:n], PGP_MODE_WEIGHTED, w.gate_deg)
        assert np.allclose(sw, swo[:n], rtol=0, atol=2e-6)
        assert bi == (int(np.argmax(sw)) if sw.max() > 0 else -1)


def test_queries_around_the_grid_boundary():
    """The cell lookup clamps instead of testing validity: queries just inside, on and far outside
    the padded grid (incl. the scene's extreme points, NaN and huge coordinates) must count exactly
    as the oracle does."""
    rng = np.random.default_rng(123)
    delta = np.float32(0.01)
    # scene: the six faces of a box, so that extreme points sit on every side of the bounding box
    n = 3000
    P = rng.uniform(-0.2, 0.2, (n, 3)).astype(np.float32)
    face = rng.integers(0, 6, n)
    P[np.arange(n), face % 3] = np.where(face < 3, -0.2, 0.2).astype(np.float32)
    Pn = np.zeros_like(P)
    Pn[np.arange(n), face % 3] = np.where(face < 3, -1, 1)
    Pw = rng.random(n).astype(np.float32)
    # model: scene points displaced by up to 3 delta along each axis (in / on / outside the reach)
    Q = (P[rng.choice(n, 700, replace=False)] + rng.uniform(-3, 3, (700, 3)).astype(np.float32) * delta).astype(np.float32)
    Qn = np.tile(np.array([0, 0, 1], np.float32), (700, 1))
    Ts = []
    for k in range(40):
        T = np.eye(4)
        T[:3, 3] = rng.choice([-1, 0, 1], 3) * rng.choice([0.0, 0.5, 1.0, 1.02, 2.0, 40.0]) * float(delta)
        Ts.append(synth.colmajor16(T))
    T = np.eye(4); T[:3, 3] = 1e30; Ts.append(synth.colmajor16(T))
    T = np.eye(4); T[0, 3] = np.nan; Ts.append(synth.colmajor16(T))
    T = np.eye(4); T[1, 1] = np.inf; Ts.append(synth.colmajor16(T))
    Ts = np.stack(Ts).astype(np.float32)
    sc = LcpScorer()
    sc.init(P, Pn, Pw, Q, Qn, float(delta))
    orc = Oracle(P, Pn, Pw, Q, Qn, use_kd=False)
    s, c, bi, _ = sc.score(Ts)
    so, bio, _ = orc.score_batch(Ts, float(delta), mode=0)
    assert np.array_equal(s, so) and bi == bio
    sw = sc.score(Ts, PGP_MODE_WEIGHTED, 30.0)[0]
    swo, _, _ = orc.score_batch(Ts, float(delta), mode=1, gate_deg=30.0)
    assert np.allclose(sw, swo, rtol=0, atol=2e-6)
    assert s[-3:].max() == 0 and sw[-3:].max() == 0
