"""Several (segment, target) pairs refined by ONE launch (pgp_icp_refine_multi_device, csrc/icp.hip icp_persist_multi)
and the device-side hand-off of the best hypotheses (pgp_select_top_device, csrc/select.hip): the children of an MCTS
expansion belong to different objects (UCTSearch.cpp:200-266 -> UCTState.cpp:121-204), the node's object loop refines
every object's candidates (SceneCfg.cpp:379-402).  Results must equal one pgp_icp_refine_device call per job, bit for
bit, and the host-pointer call."""
import numpy as np
import pytest
import torch

from physimglobalpose_amd import LcpScorer, synth
from test_icp_index_gpu import _problem

pytestmark = pytest.mark.gpu


def _dev4(x):
    d = torch.zeros(len(x), 4, device="cuda")
    d[:, :3] = torch.from_numpy(np.ascontiguousarray(x, np.float32)).cuda()
    return d


@pytest.mark.parametrize("shapes", [((3000, 1700, 24), (5000, 2500, 64), (2200, 900, 40)),       # points per thread 2 / 3
                                    ((4000, 3500, 5), (1200, 600, 1), (2500, 1200, 0), (3000, 2000, 9))])   # 4; an empty job
def test_one_launch_equals_one_call_per_job(shapes, monkeypatch):
    probs = [_problem(60 + k, m, s, max(g, 1), rot_deg=5.0, trans=0.006, outliers=0.05) for k, (m, s, g) in enumerate(shapes)]
    scs = [LcpScorer() for _ in probs]
    jobs, ref = [], []
    for k, ((S, M, N, G), sc) in enumerate(zip(probs, scs)):
        G = G[:shapes[k][2]]
        d_src, d_tgt = _dev4(S), _dev4(M)
        d_T = torch.from_numpy(G.copy()).cuda().reshape(-1, 16)
        d_e = torch.zeros(len(G), device="cuda")
        d_it = torch.zeros(len(G), dtype=torch.int32, device="cuda")
        jobs.append(dict(scorer=sc, d_src4=d_src, d_tgt4=d_tgt, d_T=d_T, d_energy=d_e, d_iters=d_it, target_token=100 + k))
        ref.append(sc.icp_refine(S, M, G, trim=0.9, max_iterations=30) if len(G) else None)
    LcpScorer.icp_refine_multi_device(jobs, trim=0.9, max_iterations=30)
    torch.cuda.synchronize()
    for q, r in zip(jobs, ref):
        if r is None:
            continue
        assert np.array_equal(q["d_T"].cpu().numpy(), r[0]) and np.array_equal(q["d_energy"].cpu().numpy(), r[1])
        assert np.array_equal(q["d_iters"].cpu().numpy(), r[2]) and (r[2] >= 1).all()
    # job by job through the same entry (PGP_ICP_MULTI=0): the fallback gives the same bits
    for k, q in enumerate(jobs):
        q["d_T"].copy_(torch.from_numpy(probs[k][3][:shapes[k][2]].copy()).cuda().reshape(-1, 16))
    monkeypatch.setenv("PGP_ICP_MULTI", "0")
    LcpScorer.icp_refine_multi_device(jobs, trim=0.9, max_iterations=30)
    torch.cuda.synchronize()
    for q, r in zip(jobs, ref):
        if r is not None:
            assert np.array_equal(q["d_T"].cpu().numpy(), r[0])


def test_select_top_matches_a_stable_argsort_and_the_rigid_inverse():
    rng = np.random.default_rng(5)
    n, k = 5000, 64
    T = np.stack([synth.colmajor16(synth._se3(synth._random_rot(rng), rng.uniform(-1, 1, 3))) for _ in range(n)])
    s = rng.uniform(0, 1, n).astype(np.float32)
    s[rng.choice(n, 200, replace=False)] = s[7]          # many exact ties (also at the top)
    s[7] = s.max() if False else s[7]
    s[rng.choice(n, 300, replace=False)] = 0.0           # never selected
    s[11] = np.nan
    sc = LcpScorer()
    d_T, d_s = torch.from_numpy(T).cuda(), torch.from_numpy(s).cuda()
    for invert in (False, True):
        d_out, d_idx, d_n = sc.select_top_device(d_T, d_s, k, invert=invert)
        torch.cuda.synchronize()
        key = np.where(np.isfinite(s) & (s > 0), s, 0.0)
        order = np.argsort(-key, kind="stable")[:k]
        assert int(d_n[0]) == k and np.array_equal(d_idx.cpu().numpy(), order)
        got = d_out.cpu().numpy()
        if not invert:
            assert np.array_equal(got, T[order])
        else:
            for g, h in zip(got, order):
                M = T[h].reshape(4, 4, order="F").astype(np.float64)
                inv = np.eye(4)
                inv[:3, :3] = M[:3, :3].T
                inv[:3, 3] = -(M[:3, :3].T @ M[:3, 3])
                assert np.array_equal(g, synth.colmajor16(inv).astype(np.float32)) or np.abs(g - synth.colmajor16(inv)).max() < 1e-7
    # fewer positive scores than k: the tail is marked
    s2 = np.zeros(n, np.float32)
    s2[[5, 900, 17]] = [0.3, 0.9, 0.3]
    d_out, d_idx, d_n = sc.select_top_device(d_T, torch.from_numpy(s2).cuda(), 8, invert=False)
    torch.cuda.synchronize()
    assert int(d_n[0]) == 3 and d_idx.cpu().numpy().tolist() == [900, 5, 17, -1, -1, -1, -1, -1]
    assert np.array_equal(d_out[3:].cpu().numpy(), np.tile(np.eye(4, dtype=np.float32).reshape(16), (5, 1)))


def test_two_jobs_on_one_context_run_one_after_the_other():
    """A context keeps ONE target index: two jobs that name the same scorer (an object with two segments, two targets on
    one context) must not share a launch whose second index build overwrites the first -- same bits as one
    call per job."""
    probs = [_problem(90, 3000, 1500, 12, rot_deg=4.0, trans=0.005, outliers=0.05),
             _problem(91, 4200, 2100, 20, rot_deg=4.0, trans=0.005, outliers=0.05),
             _problem(92, 2500, 1000, 7, rot_deg=4.0, trans=0.005, outliers=0.05)]
    shared, own = LcpScorer(), LcpScorer()
    scs = [shared, own, shared]                       # jobs 0 and 2: different targets, one context
    jobs, ref = [], []
    for (S, M, N, G), sc in zip(probs, scs):
        checker = LcpScorer()
        ref.append(checker.icp_refine(S, M, G, trim=0.9, max_iterations=30))
        jobs.append(dict(scorer=sc, d_src4=_dev4(S), d_tgt4=_dev4(M), d_T=torch.from_numpy(G.copy()).cuda().reshape(-1, 16),
                         d_energy=torch.zeros(len(G), device="cuda"), d_iters=torch.zeros(len(G), dtype=torch.int32, device="cuda")))
    LcpScorer.icp_refine_multi_device(jobs, trim=0.9, max_iterations=30)
    torch.cuda.synchronize()
    for q, r in zip(jobs, ref):
        assert np.array_equal(q["d_T"].cpu().numpy(), r[0]) and np.array_equal(q["d_energy"].cpu().numpy(), r[1])
        assert np.array_equal(q["d_iters"].cpu().numpy(), r[2])
