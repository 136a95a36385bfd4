"""Sparse form of the scene index (csrc/grid_index.hip: blocks_count / blocks_insert / blocks_finish, the hashed
table of occupied 4 x 4 x 2 blocks; csrc/lcp_score.hip: block_probe): a scene whose bounding box is mostly empty
keeps the 0.85 delta cell at any extent, and every consumer of the index (scoring in both modes, records,
neighbour counts) returns exactly what the dense block array returns and what the brute-force oracle returns
(the kd-tree of Match4PCSBase::initKdTree, match4pcsBase.cc:1046-1056, answers the same radius query)."""
import numpy as np
import pytest

from physimglobalpose_amd import LcpScorer, PGP_MODE_PLAIN, PGP_MODE_WEIGHTED, synth
from _checkers import Oracle

pytestmark = pytest.mark.gpu
W_TOL = 2e-6    # weighted scores: float sums in wave order vs the reference's sequential order (tests/test_lcp_gpu.py)
I16 = synth.colmajor16(np.eye(4))


def _room(rng, n, size=(5.0, 5.0, 2.5)):
    """Floor, two walls and a few table-top clutter blobs of a room `size` metres across."""
    sx, sy, sz = size
    k = n // 5
    floor = np.c_[rng.uniform(0, sx, k), rng.uniform(0, sy, k), 0.002 * rng.standard_normal(k)]
    wall1 = np.c_[rng.uniform(0, sx, k), 0.002 * rng.standard_normal(k) + sy, rng.uniform(0, sz, k)]
    wall2 = np.c_[0.002 * rng.standard_normal(k), rng.uniform(0, sy, k), rng.uniform(0, sz, k)]
    blobs = []
    for _ in range(8):
        c = np.array([rng.uniform(0.5, sx - 0.5), rng.uniform(0.5, sy - 0.5), rng.uniform(0.3, 1.2)])
        m = synth.make_model(rng, (n - 3 * k) // 8)[0]
        blobs.append(m + c)
    P = np.concatenate([floor, wall1, wall2] + blobs).astype(np.float32)
    Pn = synth._unit(rng.standard_normal(P.shape)).astype(np.float32)
    return P, Pn, blobs


def _hypotheses(rng, n, centre, rot_deg=3.0, trans=0.004):
    T = []
    for i in range(n):
        R = synth._random_rot(rng, np.deg2rad(rot_deg * (i % 4)))
        M = synth._se3(np.eye(3), centre) @ synth._se3(R, trans * rng.standard_normal(3)) @ synth._se3(np.eye(3), -centre)
        T.append(synth.colmajor16(M))
    return np.stack(T)


def test_five_metre_room_at_5mm_against_brute_force():
    """VERDICT r2 item 9's case: 5 m extent at delta = 5 mm needs 1177+ cells per axis at h = 0.85 delta; the dense
    form would have to grow the cell (or hold 1.6 G cells).  The sparse form keeps h and matches brute force."""
    rng = np.random.default_rng(40)
    P, Pn, blobs = _room(rng, 60000)
    Pw = rng.uniform(0.2, 1.0, len(P)).astype(np.float32)
    obj = blobs[3].astype(np.float32)
    Q = obj[rng.choice(len(obj), 700, replace=False)]
    Qn = synth._unit(rng.standard_normal(Q.shape)).astype(np.float32)
    delta = 0.005
    sc = LcpScorer()
    sc.init(P, Pn, Pw, Q, Qn, delta)
    info = sc.index_info()
    assert info["sparse"] == 1
    assert max(info["grid_nx"], info["grid_ny"]) > 1024                 # beyond the dense form's axis limit
    assert abs(info["cell_size"] - 0.85 * delta) < 1e-6                 # the cell did not grow
    assert info["n_blocks"] * 32 == info["n_cells"] and info["n_cells"] < 50e6
    T = _hypotheses(rng, 24, obj.mean(0))
    T[0] = I16                                                           # Q is a subset of P: exactly 1.0
    T[5] = synth.colmajor16(synth._se3(np.eye(3), [7.0, -3.0, 1.0]))    # out of the room: zero
    orc = Oracle(P, Pn, Pw, Q, Qn, use_kd=False)
    for mode in (PGP_MODE_PLAIN, PGP_MODE_WEIGHTED):
        s, c, bi, bs = sc.score(T, mode)
        so, bio, _ = orc.score_batch(T, delta, mode=mode)
        assert np.array_equal(s, so) if mode == PGP_MODE_PLAIN else np.allclose(s, so, rtol=0, atol=W_TOL)
        assert bi == bio and s[5] == 0.0 and (mode == PGP_MODE_WEIGHTED or s[0] == 1.0)
    # per-point registrations of one hypothesis (the records of base.cc:1722-1745)
    so, reg = orc.weighted_verify(T[2], delta)
    sw, _, _, _ = sc.score(T[2:3], PGP_MODE_WEIGHTED)
    assert abs(sw[0] - np.float32(so)) <= W_TOL
    assert np.array_equal(sc.registered(T[2], PGP_MODE_WEIGHTED), reg)


def _same_in_both_forms(monkeypatch, fn):
    monkeypatch.setenv("PGP_INDEX", "dense")
    a = fn()
    monkeypatch.setenv("PGP_INDEX", "sparse")
    b = fn()
    monkeypatch.delenv("PGP_INDEX")
    return a, b


def test_sparse_equals_dense_on_a_tabletop_scene(monkeypatch):
    w = synth.make_workload(20000, 2000, 256, config_id=2)

    def run():
        sc = LcpScorer()
        sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
        info = sc.index_info()
        out = [info["sparse"], info["n_candidates"], info["n_occupied"]]
        for mode in (PGP_MODE_PLAIN, PGP_MODE_WEIGHTED):
            s, c, bi, bs = sc.score(w.T, mode, w.gate_deg)
            out += [s, c, bi, bs]
        out.append(LcpScorer().radius_outlier_filter(w.P_xyz[:5000], w.P_nrm[:5000], 0.02, 6))
        return out

    a, b = _same_in_both_forms(monkeypatch, run)
    assert a[0] == 0 and b[0] == 1
    assert b[1] >= a[1]       # the sparse form's rounding margin covers 16 384 cells per axis: a slightly wider reach
    for x, y in zip(a[3:7], b[3:7]):                                    # plain: scores, counts, best -- identical
        assert np.array_equal(x, y)
    assert np.allclose(a[7], b[7], rtol=0, atol=W_TOL) and np.array_equal(a[8], b[8]) and a[9] == b[9]
    for x, y in zip(a[-1], b[-1]):
        assert np.array_equal(x, y)


def test_sparse_edge_scenes(monkeypatch):
    """Empty scene, one point, duplicates, a NaN point, a scene far from the origin."""
    rng = np.random.default_rng(41)
    Q = rng.uniform(-0.02, 0.02, (50, 3)).astype(np.float32)
    Qn = synth._unit(rng.standard_normal(Q.shape)).astype(np.float32)
    monkeypatch.setenv("PGP_INDEX", "sparse")
    sc = LcpScorer()
    sc.init(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), None, Q, Qn, 0.005)
    s, c, bi, bs = sc.score(np.stack([I16] * 3))
    assert sc.index_info()["sparse"] == 1 and not s.any() and bi == -1
    one = np.array([[0.001, -0.002, 0.003]], np.float32)
    sc.init(one, one, None, Q, Qn, 0.005)
    s, c, bi, bs = sc.score(np.stack([I16]))
    so, bio, _ = Oracle(one, one, np.ones(1, np.float32), Q, Qn, use_kd=False).score_batch(np.stack([I16]), 0.005)
    assert np.array_equal(s, so) and bi == bio
    P = np.concatenate([Q, Q, Q[:10] + np.float32(0.001)])
    P[7] = np.nan
    Pn = synth._unit(rng.standard_normal(P.shape)).astype(np.float32)
    for off in (0.0, 900.0):
        Po = (P + np.float32(off)).astype(np.float32)
        Qo = (Q + np.float32(off)).astype(np.float32)
        sc.init(Po, Pn, None, Qo, Qn, 0.005)
        T = _hypotheses(rng, 6, Qo.mean(0).astype(np.float64), rot_deg=1.0, trans=0.002)
        orc = Oracle(Po, Pn, np.ones(len(Po), np.float32), Qo, Qn, use_kd=False)
        for mode in (PGP_MODE_PLAIN, PGP_MODE_WEIGHTED):
            s, c, bi, bs = sc.score(T, mode)
            so, bio, _ = orc.score_batch(T, 0.005, mode=mode)
            assert (np.array_equal(s, so) if mode == PGP_MODE_PLAIN else np.allclose(s, so, rtol=0, atol=W_TOL)) and bi == bio


def test_long_flat_scene_keeps_the_cell(monkeypatch):
    """40 m x 30 m x 1 m at delta = 5 mm: 9400 x 7060 x 240 cells; only the sparse form can hold it at h = 0.85 delta."""
    rng = np.random.default_rng(42)
    P = np.c_[rng.uniform(0, 40, 30000), rng.uniform(0, 30, 30000), rng.uniform(0, 1, 30000)].astype(np.float32)
    Q = (P[rng.choice(30000, 400, replace=False)]).astype(np.float32)
    sc = LcpScorer()
    sc.init(P, None, None, Q, None, 0.005)
    info = sc.index_info()
    assert info["sparse"] == 1 and abs(info["cell_size"] - 0.00425) < 1e-6 and info["grid_nx"] > 9000
    T = np.stack([I16, synth.colmajor16(synth._se3(np.eye(3), [0.002, 0.001, -0.002])),
                  synth.colmajor16(synth._se3(np.eye(3), [0.2, 0.0, 0.0]))])
    s, c, bi, bs = sc.score(T)
    orc = Oracle(P, np.zeros_like(P), np.ones(len(P), np.float32), Q, np.zeros_like(Q), use_kd=False)
    so, bio, _ = orc.score_batch(T, 0.005, mode=PGP_MODE_PLAIN)
    assert s[0] == 1.0 and np.array_equal(s, so) and bi == bio
