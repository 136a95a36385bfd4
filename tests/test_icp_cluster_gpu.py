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
