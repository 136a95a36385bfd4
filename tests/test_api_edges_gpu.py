"""Edge cases of the secondary entry points (through the C ABI): empty and degenerate inputs,
caps, non-finite transforms.  The scoring entry points have their own file (test_edge_gpu.py)."""
import numpy as np
import pytest

from physimglobalpose_amd import LcpScorer, synth, _lib

pytestmark = pytest.mark.gpu
I16 = synth.colmajor16(np.eye(4))


def test_congruent_entry_points_on_tiny_and_empty_inputs():
    sc = LcpScorer()
    with pytest.raises(_lib.PgpError):
        sc.extract_pairs(0.1, 0.005)                       # no search model yet
    sc.set_search_model(np.zeros((1, 3), np.float32))
    assert len(sc.extract_pairs(0.1, 0.005)) == 0         # a single point has no pairs
    pts = np.array([[0, 0, 0], [0.1, 0, 0], [0, 0.1, 0], [0.1, 0.1, 0]], np.float32)
    sc.set_search_model(pts)
    p = sc.extract_pairs(0.1, 0.001)
    assert {tuple(x) for x in p.tolist()} == {(0, 1), (1, 0), (0, 2), (2, 0), (1, 3), (3, 1), (2, 3), (3, 2)}
    assert len(sc.extract_pairs(0.1, 0.001, cap=3)) == 3  # truncated, no overflow
    base = pts.copy()
    assert len(sc.find_congruent(base, 0.5, 0.5, 0.005, np.zeros((0, 2), np.int32), p)) == 0
    # crossing diagonals of the square: the two segments meet at their midpoints
    diag = sc.extract_pairs(np.float32(0.1 * np.sqrt(2)), 0.001)
    assert {tuple(x) for x in diag.tolist()} == {(0, 3), (3, 0), (1, 2), (2, 1)}
    cross = pts[[0, 3, 1, 2]]
    q = sc.find_congruent(cross, 0.5, 0.5, 0.005, diag, diag)
    # whatever the reference's cone rasterisation yields here (nothing, for this exactly planar
    # toy), the GPU yields the same list
    from _checkers import CongruentChecker
    assert np.array_equal(q, CongruentChecker(pts, "oracle").find_congruent(cross, 0.5, 0.5, 0.005, diag, diag))
    # parallel base segments: the reference's cone angle is acos(1) = 0 -> zero samples
    par = np.array([[0, 0, 0], [0.1, 0, 0], [0, 0.1, 0], [0.1, 0.1, 0]], np.float32)
    assert len(sc.find_congruent(par, 0.5, 0.5, 0.005, p, p)) >= 0
    # out-of-range pair ids are ignored, not dereferenced
    bad = np.array([[0, 99], [1, 2]], np.int32)
    sc.find_congruent(base, 0.5, 0.5, 0.005, bad, p)
    with pytest.raises(_lib.PgpError):
        sc.find_congruent(base, 0.5, 0.5, 0.0, p, p)       # threshold 0 -> unusable grid


def test_icp_with_nonfinite_guess_and_tiny_clouds():
    rng = np.random.default_rng(0)
    M = rng.uniform(-0.1, 0.1, (50, 3)).astype(np.float32)
    S = M[:7] + np.float32(0.001)
    G = np.stack([I16, I16.copy()])
    G[1, 12] = np.nan
    sc = LcpScorer()
    T, e, it = sc.icp_refine(S, M, G, trim=0.7, max_iterations=20)
    assert np.isfinite(T[0]).all() and e[0] < 1e-5 and it[0] >= 1
    assert it[1] >= 1                                       # the NaN pose terminates too
    T1, _, it1 = sc.icp_refine(M[:1], M[:1], I16[None], trim=1.0)   # one point each
    assert np.isfinite(T1).all() and it1[0] >= 1


def test_rigid_fit_empty_batch_and_state_errors():
    sc = LcpScorer()
    with pytest.raises(_lib.PgpError):
        sc.rigid_from_congruent(np.zeros((1, 4), np.int32), np.zeros((1, 4), np.int32), np.zeros(3), np.zeros(3))
    sc.set_scene(np.zeros((4, 3), np.float32), None, None, 0.005)
    sc.set_search_model(np.zeros((4, 3), np.float32))
    T, pose, st, rms = sc.rigid_from_congruent(np.zeros((0, 4), np.int32), np.zeros((0, 4), np.int32),
                                               np.zeros(3), np.zeros(3))
    assert len(T) == 0
    # all points coincide: every fit is degenerate (status 2), transforms are NaN
    T, pose, st, rms = sc.rigid_from_congruent(np.array([[0, 1, 2, 3]], np.int32), np.array([[0, 1, 2, 3]], np.int32),
                                               np.zeros(3), np.zeros(3))
    assert st[0] == 2 and np.isnan(T).all()


def test_distinct_contexts_are_safe_from_different_threads():
    """One context per call-site, not shared: two host threads, each with its own context (the
    reference's commented-out per-object threads, SceneCfg.cpp:377,404-405), score concurrently --
    ctypes drops the GIL during the calls -- and both get the oracle's scores."""
    import threading
    from physimglobalpose_amd import PGP_MODE_WEIGHTED
    from _checkers import Oracle
    ws = [synth.make_workload(5000, 700, 256, config_id=81 + k) for k in range(2)]
    want = []
    for w in ws:
        orc = Oracle(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm)
        want.append((orc.score_batch(w.T, w.delta, mode=0)[0], orc.score_batch(w.T, w.delta, mode=1, gate_deg=w.gate_deg)[0]))
    errors = []

    def work(k):
        try:
            w = ws[k]
            sc = LcpScorer(0)
            sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
            for _ in range(30):
                s = sc.score(w.T)[0]
                sw = sc.score(w.T, PGP_MODE_WEIGHTED, w.gate_deg)[0]
                assert np.array_equal(s, want[k][0])
                assert np.allclose(sw, want[k][1], rtol=0, atol=2e-6)
        except Exception as e:  # noqa: BLE001 - reported to the main thread
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_set_model_after_queued_device_scoring_does_not_overwrite_under_the_kernel():
    """A *_device call returns with its kernels queued on the caller's stream; a following
    pgp_set_model (which rewrites the model arrays on the context's own stream) must be ordered
    behind them (pgp.h: the context waits on an event).  A long queue in front of the scoring call
    makes the race certain without that ordering."""
    import torch
    from physimglobalpose_amd import PGP_MODE_PLAIN
    w = synth.make_workload(20000, 2000, 512, config_id=2)
    sc = LcpScorer(0)
    sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    sc.reserve(w.n_h)
    want, _, _, _ = sc.score(w.T, PGP_MODE_PLAIN)
    other = np.ascontiguousarray(w.Q_xyz[::-1] + np.float32(0.05))   # same size: no reallocation, no implicit sync
    dT = torch.from_numpy(w.T).cuda()
    ds = torch.zeros(w.n_h, device="cuda")
    side = torch.cuda.Stream()
    busy = torch.randn(4096, 4096, device="cuda")
    for _ in range(3):
        with torch.cuda.stream(side):
            for _ in range(20):
                busy = busy @ busy * 1e-4     # keeps the stream busy for milliseconds
            sc.score_device(dT, ds, mode=PGP_MODE_PLAIN, stream=side)
        sc.set_model(other, w.Q_nrm)          # returns once the NEW model is resident
        side.synchronize()
        assert np.array_equal(ds.cpu().numpy(), want)
        sc.set_model(w.Q_xyz, w.Q_nrm)


def test_round4_entry_points_on_empty_and_bad_arguments():
    """pgp_select_top_device, pgp_icp_refine_multi_device, pgp_congruent_batch_fit_score / _fetch: empty inputs are
    no-ops, bad arguments and missing state are error codes with a message, nothing is dereferenced."""
    import ctypes as C
    import torch
    sc = LcpScorer()
    L, h = sc._lib, sc._h
    d_T = torch.zeros(4, 16, device="cuda")
    d_s = torch.tensor([0.5, 0.0, 0.7, float("nan")], device="cuda")
    # k = 0: nothing happens; n = 0 with k > 0: all slots marked empty
    out, idx, n = sc.select_top_device(d_T, d_s, 3, invert=False)
    torch.cuda.synchronize()
    assert int(n[0]) == 2 and idx.cpu().numpy().tolist() == [2, 0, -1]
    assert L.pgp_select_top_device(h, None, None, 0, 0, 1, None, None, None, None) == 0
    empty_T, empty_s = torch.zeros(0, 16, device="cuda"), torch.zeros(0, device="cuda")
    out, idx, n = sc.select_top_device(empty_T, empty_s, 2)
    torch.cuda.synchronize()
    assert int(n[0]) == 0 and idx.cpu().numpy().tolist() == [-1, -1]
    assert L.pgp_select_top_device(h, None, None, 5, 2, 1, C.c_void_p(out.data_ptr()), None, C.c_void_p(n.data_ptr()), None) != 0
    assert b"bad argument" in L.pgp_last_error()
    # multi-target ICP: no jobs, a NULL context, a job without clouds
    prm = _lib.IcpParams(10, 0.9, 0.0, 1.0)
    assert L.pgp_icp_refine_multi_device(None, 0, C.byref(prm), None) == 0
    jobs = (_lib.IcpJob * 1)()
    jobs[0] = _lib.IcpJob(None, None, 0, None, 0, None, 1, None, None)
    assert L.pgp_icp_refine_multi_device(jobs, 1, C.byref(prm), None) != 0
    jobs[0] = _lib.IcpJob(h, None, 10, None, 10, d_T.data_ptr(), 4, None, None)
    assert L.pgp_icp_refine_multi_device(jobs, 1, C.byref(prm), None) != 0 and b"bad job" in L.pgp_last_error()
    LcpScorer.icp_refine_multi_device([])                                     # a no-op
    # a job with poses but zero-size clouds is refused by the launcher, not run
    z4 = torch.zeros(0, 4, device="cuda")
    with pytest.raises(_lib.PgpError):
        LcpScorer.icp_refine_multi_device([dict(scorer=sc, d_src4=z4, d_tgt4=z4, d_T=d_T.clone())])
    # fused fit + verification: nothing to fit is fine; fetching before any fit is a state error
    i4 = np.zeros(4, np.int32)
    f3 = np.zeros(3, np.float32)
    sco, st = np.zeros(1, np.float32), np.zeros(1, np.int32)
    bi, bs = C.c_int(7), C.c_float(3.0)
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))      # noqa: E731
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))    # noqa: E731
    assert L.pgp_congruent_batch_fit_score(h, ip(i4), ip(i4), 0, fp(f3), fp(f3), 1, C.c_float(30.0), fp(sco), ip(st),
                                           C.byref(bi), C.byref(bs)) == 0
    assert bi.value == -1 and bs.value == 0.0
    assert L.pgp_congruent_batch_fetch(h, ip(i4), 1, None, None) != 0 and b"no fits resident" in L.pgp_last_error()
    assert L.pgp_congruent_batch_fetch(h, ip(i4), 0, None, None) == 0
    assert L.pgp_congruent_batch_fit_score(h, None, None, 3, fp(f3), fp(f3), 1, C.c_float(30.0), None, None, None, None) != 0
