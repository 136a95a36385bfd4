"""The clustered ICP launch (several workgroups per pose, up to 128 poses: csrc/icp.hip) around its edges: a LOST meeting --
forced by PGP_ICP_FORCE_LOST -- must end in the same transforms (the pose's workgroup 0 goes on alone inside the launch, the
others leave: no repair launch, no flag for the caller), through the host-pointer and the device-pointer call; and the meeting
counters, which the kernel puts back to zero itself instead of a fill before every call, must hold across calls of changing
shape on one context.  The reference's consumer: the per-expansion refinement of the search (UCTState.cpp:121-204)."""
import numpy as np
import pytest
import torch

from physimglobalpose_amd import LcpScorer
from test_icp_index_gpu import _problem

pytestmark = pytest.mark.gpu


def _dev4(x):
    d = torch.zeros(len(x), 4, device="cuda")
    d[:, :3] = torch.from_numpy(np.ascontiguousarray(x, np.float32)).cuda()
    return d


def _same(a, b):
    return all(np.array_equal(x, y) for x, y in zip(a, b))


def test_lost_meeting_is_repaired_inside_the_launch(monkeypatch):
    S, M, N, G = _problem(81, 5000, 2500, 24, rot_deg=4.0, trans=0.004, outliers=0.03)
    sc = LcpScorer()
    monkeypatch.setenv("PGP_ICP_WGS", "1")
    ref = sc.icp_refine(S, M, G, trim=0.9, max_iterations=12)
    monkeypatch.delenv("PGP_ICP_WGS")
    assert _same(sc.icp_refine(S, M, G, trim=0.9, max_iterations=12), ref)          # clustered, nothing lost
    monkeypatch.setenv("PGP_ICP_FORCE_LOST", "1")
    assert _same(sc.icp_refine(S, M, G, trim=0.9, max_iterations=12), ref)          # lost: workgroup 0 of every pose goes on alone
    d_src, d_tgt = _dev4(S), _dev4(M)
    d_T = torch.from_numpy(G.copy()).cuda().reshape(-1, 16)
    d_e = torch.zeros(len(G), device="cuda")
    d_it = torch.zeros(len(G), dtype=torch.int32, device="cuda")
    sc.icp_refine_device(d_src, d_tgt, d_T, d_e, d_it, trim=0.9, max_iterations=12)   # the same through the device-pointer call
    torch.cuda.synchronize()
    assert _same((d_T.cpu().numpy().reshape(ref[0].shape), d_e.cpu().numpy(), d_it.cpu().numpy()), ref)
    monkeypatch.delenv("PGP_ICP_FORCE_LOST")
    # the counters are back at zero after a lost call too
    assert _same(sc.icp_refine(S, M, G, trim=0.9, max_iterations=12), ref)
    assert _same(sc.icp_refine(S, M, G, trim=0.9, max_iterations=12), ref)
    sc.close()


def test_counters_hold_across_calls_of_changing_shape(monkeypatch):
    probs = [_problem(90 + k, m, s, g, rot_deg=3.0, trans=0.003, outliers=0.02)
             for k, (m, s, g) in enumerate([(5000, 2500, 64), (3000, 1700, 8), (5000, 2500, 1), (4000, 3000, 100), (5000, 2500, 64)])]
    one = LcpScorer()
    monkeypatch.setenv("PGP_ICP_WGS", "1")
    refs = [one.icp_refine(S, M, G, trim=0.9, max_iterations=8) for S, M, N, G in probs]
    monkeypatch.delenv("PGP_ICP_WGS")
    sc = LcpScorer()
    for rep in range(3):                     # the same context, shapes in rotation, every call twice
        for (S, M, N, G), ref in zip(probs, refs):
            assert _same(sc.icp_refine(S, M, G, trim=0.9, max_iterations=8), ref)
            assert _same(sc.icp_refine(S, M, G, trim=0.9, max_iterations=8), ref)
    # a device-pointer call in between
    S, M, N, G = probs[0]
    d_T = torch.from_numpy(G.copy()).cuda().reshape(-1, 16)
    sc.icp_refine_device(_dev4(S), _dev4(M), d_T, trim=0.9, max_iterations=8)
    torch.cuda.synchronize()
    assert np.array_equal(d_T.cpu().numpy().reshape(refs[0][0].shape), refs[0][0])
    assert _same(sc.icp_refine(S, M, G, trim=0.9, max_iterations=8), refs[0])
    one.close()
    sc.close()


def test_plain_and_cooperative_launch_give_the_same_bits(monkeypatch):
    """The kernels whose workgroups wait for each other go out as plain launches of a grid that fits the device (no
    cooperative queue: its existence makes the hardware scheduler time-slice the GPU between processes, csrc/icp.hip
    launch_resident); PGP_COOPERATIVE_LAUNCH=1 is the runtime's checked form.  Same transforms, energies and iteration
    counts from both: clustered launches of 1 and 24 poses, and the scene-sized capped form in one launch."""
    S, M, N, G = _problem(83, 5000, 2500, 24, rot_deg=4.0, trans=0.004, outliers=0.03)
    rng = np.random.default_rng(12)
    tgt = np.c_[rng.uniform(-0.6, 0.6, 60000), rng.uniform(-0.4, 0.4, 60000), 0.0005 * rng.standard_normal(60000)].astype(np.float32)
    src = (tgt[rng.choice(len(tgt), 20000, replace=False)] + np.array([0.004, -0.003, 0.002]) + 0.0008 * rng.standard_normal((20000, 3))).astype(np.float32)
    eye = np.eye(4, dtype=np.float32).T.reshape(1, 16).copy()
    sc = LcpScorer()

    def run():
        return (sc.icp_refine(S, M, G, trim=0.9, max_iterations=12), sc.icp_refine(S, M, G[:1], trim=0.9, max_iterations=12),
                sc.icp_refine_ex(src, tgt, eye, max_iterations=30, max_corr_dist=0.01, energy_ratio=0.0, transformation_epsilon=1e-9,
                                 absolute_mse=1e-12))
    plain = run()
    monkeypatch.setenv("PGP_COOPERATIVE_LAUNCH", "1")
    coop = run()
    monkeypatch.delenv("PGP_COOPERATIVE_LAUNCH")
    again = run()
    for a, b, c in zip(plain, coop, again):
        assert _same(a, b) and _same(a, c)
    assert plain[2][2][0] > 3          # the scene-sized form really iterated
    sc.close()


def test_clustered_launch_behind_a_kernel_that_holds_the_device():
    """A plain launch promises no co-residency: with another stream's long kernel on the compute units the workgroups of a
    clustered launch arrive late, one after the other, and wait for their partners meanwhile.  Same bits as on an idle device."""
    S, M, N, G = _problem(85, 5000, 2500, 48, rot_deg=4.0, trans=0.004, outliers=0.03)
    sc = LcpScorer()
    ref = sc.icp_refine(S, M, G, trim=0.9, max_iterations=12)
    d_src, d_tgt = _dev4(S), _dev4(M)
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device="cuda")
    torch.cuda.synchronize()
    for rep in range(3):
        d_T = torch.from_numpy(G.copy()).cuda().reshape(-1, 16)
        d_e = torch.zeros(len(G), device="cuda")
        d_it = torch.zeros(len(G), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            b = a
            for _ in range(6):               # a few milliseconds of full-device work
                b = torch.sin(b @ a * 1e-3)
        sc.icp_refine_device(d_src, d_tgt, d_T, d_e, d_it, trim=0.9, max_iterations=12)
        torch.cuda.synchronize()
        assert _same((d_T.cpu().numpy().reshape(ref[0].shape), d_e.cpu().numpy(), d_it.cpu().numpy()), ref)
    sc.close()


def test_scene_sized_form_that_loses_a_pose_is_redone_host_driven(monkeypatch):
    """The scene-sized capped ICP in one launch ends a pose whose units do not arrive within its clock bound with
    iteration count -1 (a device held by somebody else for seconds; forced here by PGP_ICP_FORCE_LOST).  The host-pointer
    call then redoes the job with the host-driven iterations -- the checker of that kernel, same bits; the device-pointer
    call reports the -1."""
    rng = np.random.default_rng(14)
    tgt = np.c_[rng.uniform(-0.6, 0.6, 60000), rng.uniform(-0.4, 0.4, 60000), 0.0005 * rng.standard_normal(60000)].astype(np.float32)
    src = (tgt[rng.choice(len(tgt), 20000, replace=False)] + np.array([0.004, -0.003, 0.002]) + 0.0008 * rng.standard_normal((20000, 3))).astype(np.float32)
    eye = np.eye(4, dtype=np.float32).T.reshape(1, 16).copy()
    kw = dict(max_iterations=30, max_corr_dist=0.01, energy_ratio=0.0, transformation_epsilon=1e-9, absolute_mse=1e-12)
    sc = LcpScorer()
    ref = sc.icp_refine_ex(src, tgt, eye, **kw)
    assert ref[2][0] > 3
    monkeypatch.setenv("PGP_ICP_FORCE_LOST", "1")
    assert _same(sc.icp_refine_ex(src, tgt, eye, **kw), ref)
    d_T = torch.from_numpy(eye.copy()).cuda()
    d_e = torch.zeros(1, device="cuda")
    d_it = torch.zeros(1, dtype=torch.int32, device="cuda")
    sc.icp_refine_device(_dev4(src), _dev4(tgt), d_T, d_e, d_it, trim=1.0, max_iterations=30, max_corr_dist=0.01, energy_ratio=0.0)
    torch.cuda.synchronize()
    assert int(d_it[0]) == -1
    monkeypatch.delenv("PGP_ICP_FORCE_LOST")
    d_T.copy_(torch.from_numpy(eye))
    sc.icp_refine_device(_dev4(src), _dev4(tgt), d_T, d_e, d_it, trim=1.0, max_iterations=30, max_corr_dist=0.01, energy_ratio=0.0)
    torch.cuda.synchronize()
    assert int(d_it[0]) > 3
    assert _same(sc.icp_refine_ex(src, tgt, eye, **kw), ref)
    sc.close()


@pytest.mark.parametrize("n_jobs", [1, 2])
def test_group_icp_redoes_a_lost_scene_sized_piece(n_jobs, monkeypatch):
    """pgp_multi_icp_refine is a host-pointer entry too: a piece whose pose the one-launch scene-sized form gave up
    on (iters -1, transform untouched) is redone host-driven, so the caller gets one pgp_icp_refine per job -- for a member
    with a single piece (plain launch_icp) and with several (launch_icp_multi's job-by-job fall-back for big segments)."""
    from physimglobalpose_amd import MultiGpuScorer
    rng = np.random.default_rng(14)
    tgt = np.c_[rng.uniform(-0.6, 0.6, 60000), rng.uniform(-0.4, 0.4, 60000), 0.0005 * rng.standard_normal(60000)].astype(np.float32)
    src = (tgt[rng.choice(len(tgt), 20000, replace=False)] + np.array([0.004, -0.003, 0.002]) + 0.0008 * rng.standard_normal((20000, 3))).astype(np.float32)
    eye = np.eye(4, dtype=np.float32).T.reshape(1, 16).copy()
    jobs = [(src, tgt, eye)]
    if n_jobs == 2:
        S, M, N, G = _problem(85, 5000, 2500, 6, rot_deg=2.0, trans=0.002, outliers=0.0)
        jobs.append((S, M, G))
    kw = dict(trim=1.0, max_iterations=30, max_corr_dist=0.01, energy_ratio=0.0)
    sc = LcpScorer()
    ref = [sc.icp_refine(s, t, g, **kw) for s, t, g in jobs]
    assert ref[0][2][0] > 3
    monkeypatch.delenv("PGP_MULTI_EMULATE", raising=False)
    grp = MultiGpuScorer([0])
    monkeypatch.setenv("PGP_ICP_FORCE_LOST", "1")
    # (the knob reaches the form: the device-pointer call reports the lost pose)
    d_T, d_it = torch.from_numpy(eye.copy()).cuda(), torch.zeros(1, dtype=torch.int32, device="cuda")
    sc.icp_refine_device(_dev4(src), _dev4(tgt), d_T, None, d_it, **kw)
    torch.cuda.synchronize()
    assert int(d_it[0]) == -1
    got = grp.icp_refine(jobs, **kw)
    for a, b in zip(got, ref):
        assert _same(a, b)
    monkeypatch.delenv("PGP_ICP_FORCE_LOST")
    got = grp.icp_refine(jobs, **kw)
    for a, b in zip(got, ref):
        assert _same(a, b)
    grp.close()
    sc.close()


@pytest.mark.parametrize("wait_ms", ["0.002", "0.02", "0.2"])
def test_meetings_that_run_out_at_random_keep_the_bits(wait_ms, monkeypatch):
    """The waits' clock bounds follow the work now (3 ms floor; 2 s up to round 5).  With the floor pulled down to 0.2 .. 20 us
    (PGP_ICP_WAIT_MS) meetings run out for real, at random points of random poses: the first workgroup to notice marks the
    pose abandoned, its partners leave at once, workgroup 0 goes on alone -- and the scene-sized form reports the pose lost and
    the host-pointer call redoes it.  Whatever happens when: the undisturbed launch's bits."""
    S, M, N, G = _problem(86, 5000, 2500, 64, rot_deg=4.0, trans=0.004, outliers=0.03)
    rng = np.random.default_rng(15)
    tgt = np.c_[rng.uniform(-0.6, 0.6, 60000), rng.uniform(-0.4, 0.4, 60000), 0.0005 * rng.standard_normal(60000)].astype(np.float32)
    src = (tgt[rng.choice(len(tgt), 20000, replace=False)] + np.array([0.004, -0.003, 0.002]) + 0.0008 * rng.standard_normal((20000, 3))).astype(np.float32)
    eye = np.eye(4, dtype=np.float32).T.reshape(1, 16).copy()
    kw = dict(max_iterations=30, max_corr_dist=0.01, energy_ratio=0.0, transformation_epsilon=1e-9, absolute_mse=1e-12)
    sc = LcpScorer()
    monkeypatch.delenv("PGP_ICP_WAIT_MS", raising=False)
    ref_c = [sc.icp_refine(S, M, G[:n], trim=0.9, max_iterations=12) for n in (64, 24, 7)]
    ref_s = sc.icp_refine_ex(src, tgt, eye, **kw)
    monkeypatch.setenv("PGP_ICP_WAIT_MS", wait_ms)
    for rep in range(4):
        for n, ref in zip((64, 24, 7), ref_c):
            assert _same(sc.icp_refine(S, M, G[:n], trim=0.9, max_iterations=12), ref), (wait_ms, rep, n)
        assert _same(sc.icp_refine_ex(src, tgt, eye, **kw), ref_s), (wait_ms, rep)
    monkeypatch.delenv("PGP_ICP_WAIT_MS")
    assert _same(sc.icp_refine(S, M, G, trim=0.9, max_iterations=12), ref_c[0])      # the counters were left clean
    sc.close()
