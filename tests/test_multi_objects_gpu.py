"""The native device group beyond the one LCP call (SURVEY 8e line 4, BASELINE configs[3]; csrc/multi_gpu.hip):
  * several OBJECTS in one group -- the flat (object, hypothesis) space sliced over the members, ONE exchange for the
    concatenated {scores | counts}, per-object arg-max with settlement on member 0 (SceneCfg.cpp:376-406);
  * ICP pose shards -- the (job, pose) space sliced the same way, every member one launch, results gathered
    (UCTSearch.cpp:200-266 -> UCTState.cpp:121-204);
  * congruent sets sharded by base, picks routed to the member that owns the base (base.cc:1855-1874).
Everything through the C ABI (pgp_multi_*), on ONE device with PGP_MULTI_EMULATE=n members; every result must equal the
single-context calls bit for bit.  tests/test_multi_hw_gpu.py repeats the cases over physical devices."""
import os
import tempfile

import numpy as np
import pytest

from physimglobalpose_amd import LcpScorer, MultiGpuScorer, PGP_MODE_PLAIN, PGP_MODE_WEIGHTED, synth
from _checkers import Oracle
from _dropin import make_dropin_case
from test_icp_index_gpu import _problem

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _group(monkeypatch, n, n_objects=1):
    monkeypatch.setenv("PGP_MULTI_EMULATE", str(n))
    grp = MultiGpuScorer([0])
    assert grp.n_devices == n
    for _ in range(n_objects - 1):
        grp.add_object()
    assert grp.n_objects == n_objects
    return grp


def test_flat_slices_cover_the_space_like_the_python_twin():
    from physimglobalpose_amd.sharding import flat_slices
    for counts in ([16384, 12288, 12288, 8192, 8192, 8192], [5, 0, 3], [0, 0], [1], [7, 1, 1, 1, 90]):
        for n in (1, 2, 3, 8):
            seen = []
            for k in range(n):
                got = MultiGpuScorer.flat_slices(counts, k, n)
                assert got == flat_slices(counts, k, n)
                seen += [(o, i) for o, lo, hi in got for i in range(lo, hi)]
            assert seen == [(o, i) for o, c in enumerate(counts) for i in range(c)]


def check_objects_equal_single_contexts(grp, objs, counts, modes=(PGP_MODE_PLAIN, PGP_MODE_WEIGHTED)):
    ones = []
    for o, w in enumerate(objs):
        grp.init_object(o, w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
        sc = LcpScorer(0)
        sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
        ones.append(sc)
    for mode in modes:
        Ts = [w.T[:c] for w, c in zip(objs, counts)]
        got = grp.score_objects(Ts, mode, objs[0].gate_deg)
        for o, (sc, T) in enumerate(zip(ones, Ts)):
            a = sc.score(T, mode, objs[0].gate_deg)
            b = got[o]
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2:] == b[2:], (mode, o)
        grp.upload_objects(Ts)
        again = grp.score_objects_uploaded(mode, objs[0].gate_deg)
        for a, b in zip(got, again):
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2:] == b[2:]
    return ones


@pytest.mark.parametrize("n", [2, 3, 8])
def test_objects_in_one_group_equal_single_contexts(n, monkeypatch):
    """different cloud sizes per object, lists shorter than the group, an empty list, slices that cut through objects"""
    shapes = [(9000, 1500, 700), (4000, 800, 0), (12000, 2000, 333), (3000, 500, 2), (6000, 1000, 1201)]
    objs = [synth.make_workload(p, q, max(h, 1), config_id=500 + k) for k, (p, q, h) in enumerate(shapes)]
    grp = _group(monkeypatch, n, len(objs))
    check_objects_equal_single_contexts(grp, objs, [h for _, _, h in shapes])
    # fewer lists than objects: the first objects of the group
    got = grp.score_objects([objs[0].T[:50], objs[1].T[:1]], PGP_MODE_WEIGHTED, 30.0)
    assert len(got) == 2 and len(got[0][0]) == 50 and len(got[1][0]) == 1
    # the single-object entry points still act on object 0
    s, c, bi, bs = grp.score(objs[0].T[:64], PGP_MODE_WEIGHTED, 30.0)
    assert np.array_equal(s[:50], got[0][0])
    grp.close()


def test_six_objects_64k_hypotheses_through_the_c_abi(monkeypatch):
    """BASELINE.json configs[3] at full size -- 6 objects (20 000-point segments, 3 000-point models), 65 536 hypotheses --
    as a group of 8 members through pgp_multi_score_objects: equal to six single contexts bit for bit, oracle on a sample."""
    counts = [16384, 12288, 12288, 8192, 8192, 8192]
    objs = [synth.make_workload(20000, 3000, c, config_id=300 + k) for k, c in enumerate(counts)]
    grp = _group(monkeypatch, 8, 6)
    pieces = [MultiGpuScorer.flat_slices(counts, k, 8) for k in range(8)]
    assert all(sum(hi - lo for _, lo, hi in p) == 8192 for p in pieces)
    assert any(len(p) > 1 for p in pieces)                     # a member's share cuts through an object boundary
    ones = check_objects_equal_single_contexts(grp, objs, counts, modes=(PGP_MODE_WEIGHTED,))
    got = grp.score_objects([w.T for w in objs], PGP_MODE_WEIGHTED, 30.0)
    for w, (s, c, bi, bs) in zip(objs, got):
        idx = np.unique(np.concatenate([np.arange(0, len(s), 1021), [bi]]))
        orc = Oracle(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm)
        so, _, _ = orc.score_batch(w.T[idx], w.delta, mode=1, gate_deg=w.gate_deg, threads=8)
        assert np.allclose(s[idx], so, rtol=0, atol=2e-6)
        assert bi == int(np.argmax(s)) and np.float32(bs) == s.max()
    grp.close()


@pytest.mark.parametrize("n", [2, 8])
def test_near_tie_settlement_and_exact_records_per_object(n, monkeypatch):
    """object 1 carries the near-tie fixture (its cluster straddles members), object 0 and 2 ordinary batches: the
    per-object arg-max is settled on member 0 in the reference's summation order; exact records per object"""
    g = np.load(os.path.join(GOLD, "near_ties.npz"))
    wa = synth.make_workload(6000, 900, 301, config_id=71)
    wb = synth.make_workload(5000, 700, 1000, config_id=72)
    grp = _group(monkeypatch, n, 3)
    grp.init_object(0, wa.P_xyz, wa.P_nrm, wa.P_w, wa.Q_xyz, wa.Q_nrm, wa.delta)
    grp.init_object(1, g["P"], g["Pn"], g["Pw"], g["Q"], g["Qn"], float(g["delta"]))
    grp.init_object(2, wb.P_xyz, wb.P_nrm, wb.P_w, wb.Q_xyz, wb.Q_nrm, wb.delta)
    got = grp.score_objects([wa.T, g["T"], wb.T], PGP_MODE_WEIGHTED, 30.0)
    s, c, bi, bs = got[1]
    assert bi == int(g["best_weighted"]) and np.float32(bs) == g["wscores"][bi]
    assert np.allclose(s, g["wscores"], rtol=0, atol=2e-6)
    # exact records on object 2 only (member 0's context of that object)
    from physimglobalpose_amd import _lib
    _lib.check(_lib.load().pgp_set_exact_records(grp.object_context(2, 0), 1))
    one = LcpScorer(0)
    one.init(wb.P_xyz, wb.P_nrm, wb.P_w, wb.Q_xyz, wb.Q_nrm, wb.delta)
    s0, _, bi0, _ = one.score(wb.T, PGP_MODE_WEIGHTED, 30.0)
    rng = np.random.default_rng(11)
    base = wb.T[bi0].reshape(4, 4, order="F").astype(np.float64)
    crowd = np.stack([synth.colmajor16(synth._se3(synth._random_rot(rng, 2e-4), 2e-5 * rng.standard_normal(3)) @ base)
                      for _ in range(800)])
    T2 = np.concatenate([wb.T, crowd])[rng.permutation(1800)]
    one.set_exact_records(True)
    a = one.score(T2, PGP_MODE_WEIGHTED, 30.0)
    b = grp.score_objects([wa.T, g["T"], T2], PGP_MODE_WEIGHTED, 30.0)[2]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2:] == b[2:]
    grp.close()


def check_icp_shards_equal_single_calls(grp):
    shapes = [(3000, 1700, 37), (5000, 2500, 64), (2200, 900, 5), (2600, 1300, 0), (4000, 3500, 21)]
    probs = [_problem(160 + k, m, s, max(g, 1), rot_deg=5.0, trans=0.006, outliers=0.05) for k, (m, s, g) in enumerate(shapes)]
    jobs = [(S, M, G[:shapes[k][2]]) for k, (S, M, N, G) in enumerate(probs)]
    ref = []
    for S, M, G in jobs:
        sc = LcpScorer(0)
        ref.append(sc.icp_refine(S, M, G, trim=0.9, max_iterations=30) if len(G) else None)
    for rep in range(2):                      # the second call finds every target's index resident
        got = grp.icp_refine(jobs, trim=0.9, max_iterations=30)
        for j, (r, q) in enumerate(zip(ref, got)):
            if r is None:
                assert len(q[0]) == 0
                continue
            assert np.array_equal(q[0], r[0]), (rep, j)
            assert np.array_equal(q[1], r[1]) and np.array_equal(q[2], r[2]) and (r[2] >= 1).all()
    # one job, fewer poses than members
    S, M, G = jobs[0]
    got = grp.icp_refine([(S, M, G[:1])], trim=0.9, max_iterations=30)
    assert np.array_equal(got[0][0], ref[0][0][:1])


@pytest.mark.parametrize("n", [2, 3, 8])
def test_icp_pose_shards_equal_single_calls(n, monkeypatch):
    grp = _group(monkeypatch, n)
    check_icp_shards_equal_single_calls(grp)
    grp.close()


@pytest.fixture(scope="module")
def congruent_case():
    with tempfile.TemporaryDirectory() as d:
        _, c = make_dropin_case(d, n_scene=8000, n_model=1500, n_search=500)
    w, table = c["w"], c["table"]
    keys = np.array(list(table.keys()), np.int32)
    counts = np.array([len(table[tuple(k)]) for k in keys.tolist()], np.int32)
    pairs = np.concatenate([np.array(table[tuple(k)], np.int32).reshape(-1, 2) for k in keys.tolist()])
    sc = LcpScorer()
    sc.set_scene(w.P_xyz, w.P_nrm, w.P_w, w.delta)
    sc.set_search_model(w.Qs_xyz)
    sc.set_ppf_map(keys, counts, pairs)
    rng = np.random.default_rng(3)
    ids, inv, status = sc.select_bases(rng.random((100, 4)))
    ok = status == 1
    return w, keys, counts, pairs, sc, ids[ok], inv[ok]


def check_congruent_shards_equal_single_context(grp, case, obj):
    w, keys, counts, pairs, sc, ids, inv = case
    assert len(ids) >= 16
    base_xyz = w.P_xyz[ids]
    n_quads = sc.find_congruent_batch(ids, base_xyz, inv, w.delta)
    picks = np.array([(b, j) for b in range(len(ids)) for j in range(n_quads[b])], np.int32).reshape(-1, 2)
    assert len(picks) > 0
    quads = sc.congruent_batch_quads(picks)
    rng = np.random.default_rng(5)
    sel = picks[rng.choice(len(picks), min(len(picks), 3000), replace=False)]
    fit = sc.congruent_batch_fit(sel, ids, w.centroid_P, w.centroid_Q)
    grp.init_object(obj, w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    grp.set_object_search_model(obj, w.Qs_xyz)
    grp.set_object_ppf_map(obj, keys, counts, pairs)
    nq = grp.find_congruent_batch(obj, ids, base_xyz, inv, w.delta)
    assert np.array_equal(nq, n_quads)
    assert np.array_equal(grp.congruent_batch_quads(obj, picks), quads)
    got = grp.congruent_batch_fit(obj, sel, ids, w.centroid_P, w.centroid_Q)
    good = fit[2] == 1
    assert np.array_equal(got[2], fit[2]) and good.any()
    assert np.array_equal(got[0][good], fit[0][good]) and np.array_equal(got[1][good], fit[1][good])
    assert np.array_equal(got[3], fit[3])
    # fewer bases than members, and none
    few = grp.find_congruent_batch(obj, ids[:3], base_xyz[:3], inv[:3], w.delta)
    assert np.array_equal(few, n_quads[:3])
    p3 = picks[picks[:, 0] < 3]
    if len(p3):
        assert np.array_equal(grp.congruent_batch_quads(obj, p3), quads[picks[:, 0] < 3])
    assert len(grp.find_congruent_batch(obj, ids[:0], base_xyz[:0], inv[:0], w.delta)) == 0


@pytest.mark.parametrize("n", [2, 3, 8])
def test_congruent_sets_sharded_by_base(n, congruent_case, monkeypatch):
    grp = _group(monkeypatch, n, 2)
    check_congruent_shards_equal_single_context(grp, congruent_case, obj=1)
    from physimglobalpose_amd._lib import PgpError
    with pytest.raises(PgpError):                      # object 0 holds no batch
        grp.congruent_batch_quads(0, np.array([[0, 0]], np.int32))
    grp.close()


def test_error_paths_of_the_group_entry_points(monkeypatch):
    """bad arguments and wrong states come back as error codes with a message (the node must never be taken down by a call)"""
    import ctypes as C
    from physimglobalpose_amd import _lib
    from physimglobalpose_amd._lib import PgpError
    L = _lib.load()
    grp = _group(monkeypatch, 2, 2)
    w = synth.make_workload(3000, 500, 40, config_id=81)
    grp.init_object(0, w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    grp.init_object(1, w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    with pytest.raises(PgpError):                                   # more lists than objects
        grp.score_objects([w.T, w.T, w.T], PGP_MODE_PLAIN)
    with pytest.raises(PgpError):                                   # an object that does not exist
        grp.init_object(7, w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    grp.upload_objects([w.T, w.T[:3]])
    with pytest.raises(PgpError):                                   # the single-object call on a two-object batch
        grp.score_uploaded(PGP_MODE_PLAIN)
    a = grp.score_objects_uploaded(PGP_MODE_PLAIN)
    assert len(a) == 2 and len(a[1][0]) == 3
    grp.upload(w.T[:5])                                             # ... and back to one object
    assert len(grp.score_uploaded(PGP_MODE_PLAIN)[0]) == 5
    # an object without clouds: the scoring call says which member failed, and the group stays usable
    grp.add_object()
    with pytest.raises(PgpError):
        grp.score_objects([w.T, w.T, w.T], PGP_MODE_PLAIN)
    assert len(grp.score_objects([w.T, w.T], PGP_MODE_WEIGHTED, 30.0)) == 2
    # ICP: nothing to do is not an error; a job with poses and no clouds is
    assert grp.icp_refine([]) == []
    arr = (_lib.MultiIcpJob * 1)()
    arr[0] = _lib.MultiIcpJob(None, 0, None, 0, None, 3, None, None)
    prm = _lib.IcpParams(10, 0.9, 0.0, 1.0)
    assert L.pgp_multi_icp_refine(grp._h, arr, 1, C.byref(prm)) == -1 and b"bad job" in L.pgp_last_error()
    # congruent sets: picks before a batch, a pick beyond the bases
    with pytest.raises(PgpError):
        grp.congruent_batch_quads(0, np.array([[0, 0]], np.int32))
    lo, hi, n = (C.c_int * 4)(), (C.c_int * 4)(), C.c_int(0)
    assert L.pgp_multi_flat_slices(None, 1, 0, 2, None, lo, hi, C.byref(n)) == -1
    cnt = (C.c_int * 2)(5, -1)
    ob = (C.c_int * 2)()
    assert L.pgp_multi_flat_slices(cnt, 2, 0, 2, ob, lo, hi, C.byref(n)) == -1      # a negative count
    grp.close()


def test_icp_pose_shards_with_more_jobs_than_context_slots(monkeypatch):
    """beyond 32 jobs the members' ICP contexts are taken by piece, not by job: 40 small jobs, two targets alternating, so a
    slot meets a different target from one piece to the next -- still the bits of one pgp_icp_refine per job"""
    grp = _group(monkeypatch, 3)
    base = [_problem(300 + k, 1200, 500, 3, rot_deg=3.0, trans=0.004) for k in range(2)]
    rng = np.random.default_rng(9)
    jobs = []
    for j in range(40):
        S, M, N, G = base[j % 2]
        jobs.append((S, M, G[: 1 + j % 3] + np.float32(0)))
    one = LcpScorer(0)
    ref = [one.icp_refine(S, M, G, trim=0.9, max_iterations=20) for S, M, G in jobs]
    for rep in range(2):
        got = grp.icp_refine(jobs, trim=0.9, max_iterations=20)
        for j, (r, q) in enumerate(zip(ref, got)):
            assert all(np.array_equal(x, y) for x, y in zip(r, q)), (rep, j)
    grp.close()
