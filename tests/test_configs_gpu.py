"""BASELINE.json configs[2] in miniature ("3 objects in clutter, 16k hypotheses/object, ICP refine,
1 MI355X") driven through the C ABI: three independent (scene segment, model) contexts on one
GPU, a hypothesis batch per object scored with the live weighted mode, the top-k refined by the
batched ICP kernel, refined poses re-scored.  Sizes are reduced so the oracle check stays in
seconds; the flow and the entry points are the full-size ones."""
import numpy as np
import pytest

from physimglobalpose_amd import LcpScorer, PGP_MODE_PLAIN, PGP_MODE_WEIGHTED, synth
from _checkers import Oracle

pytestmark = pytest.mark.gpu


def _inv16(T16):
    return synth.colmajor16(np.linalg.inv(np.asarray(T16, np.float64).reshape(4, 4).T))


def test_three_objects_score_then_icp_refine():
    objs = [synth.make_workload(12000, 1500, 2048, config_id=200 + k) for k in range(3)]
    scorers = []
    for w in objs:                                   # one context per object (SceneCfg.cpp:379-402)
        sc = LcpScorer()
        sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
        scorers.append(sc)
    for w, sc in zip(objs, scorers):
        s, c, bi, bs = sc.score(w.T, PGP_MODE_WEIGHTED, w.gate_deg)
        orc = Oracle(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm)
        idx = np.unique(np.concatenate([np.arange(0, 2048, 97), [bi]]))
        so, _, _ = orc.score_batch(w.T[idx], w.delta, mode=1, gate_deg=w.gate_deg, threads=8)
        assert np.allclose(s[idx], so, rtol=0, atol=2e-6)
        # refine the 16 best hypotheses: source = object part of the segment, target = model
        # (UCTState::performTrICP: tform = inverse(pose), align(segment -> model), invert back)
        top = np.argsort(-s, kind="stable")[:16]
        seg = w.P_xyz[w.P_w == 1.0]
        G = np.stack([_inv16(w.T[h]) for h in top])
        Gr, energy, iters = sc.icp_refine(seg, w.Q_xyz, G, trim=0.9, max_iterations=30)
        refined = np.stack([_inv16(g) for g in Gr])
        s_ref = sc.score(refined, PGP_MODE_PLAIN)[0]
        s_before = sc.score(w.T[top], PGP_MODE_PLAIN)[0]
        # trimmed ICP minimises the distance energy, not the inlier count: the refined poses agree
        # with each other and sit at the level of the best candidate (within a few inliers)
        assert s_ref.max() >= 0.97 * s_before.max()
        assert np.median(s_ref) >= np.median(s_before)
        assert s_ref.max() - s_ref.min() <= 5.0 / len(w.Q_xyz)
        assert (iters >= 1).all() and np.isfinite(energy).all()


def test_six_objects_64k_hypotheses_sharded_like_8_ranks():
    """BASELINE.json configs[3] at full hypothesis count on ONE GPU: 6 objects, 65 536 hypotheses in
    total, partitioned exactly as 8 ranks would (flat_slices), each emulated rank filling only its
    slice of the zeroed score vector through LcpScorer.score_device; the sum over ranks is what the
    all-reduce(SUM) produces and must equal the unsharded scores bit for bit.  Oracle on a sample."""
    import torch
    from physimglobalpose_amd.sharding import MultiObjectShardedScorer, best_of, flat_slices
    counts = [16384, 12288, 12288, 8192, 8192, 8192]
    assert sum(counts) == 65536
    objs = [synth.make_workload(20000, 3000, n, config_id=300 + k) for k, n in enumerate(counts)]
    scorers, Ts = [], []
    for w in objs:
        sc = LcpScorer(0)
        sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
        sc.reserve(max(counts))
        scorers.append(sc)
        Ts.append(torch.from_numpy(w.T).cuda())

    def local(sc):
        def f(T):
            s = torch.zeros(len(T), device="cuda")
            c = torch.zeros(len(T), dtype=torch.int32, device="cuda")
            b = torch.zeros(2, dtype=torch.int32, device="cuda")
            sc.score_device(T.contiguous(), s, c, b, mode=PGP_MODE_WEIGHTED)
            torch.cuda.synchronize()
            return s
        return f

    fns = [local(sc) for sc in scorers]
    whole, bests = MultiObjectShardedScorer(fns, rank=0, world=1).score(Ts)
    acc = [torch.zeros_like(s) for s in whole]
    covered = 0
    for r in range(8):
        pieces = flat_slices(counts, r, 8)
        covered += sum(hi - lo for _, lo, hi in pieces)
        part, _ = MultiObjectShardedScorer(fns, rank=r, world=8).score(Ts)   # no process group: no exchange
        for a, p in zip(acc, part):
            a += p
    assert covered == 65536
    for o, (a, s) in enumerate(zip(acc, whole)):
        assert torch.equal(a, s), f"object {o}"
        assert best_of(a) == bests[o]
    for w, s, (bi, bs) in zip(objs, whole, bests):
        s = s.cpu().numpy()
        idx = np.unique(np.concatenate([np.arange(0, len(s), 1021), [bi]]))
        orc = Oracle(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm)
        so, _, _ = orc.score_batch(w.T[idx], w.delta, mode=1, gate_deg=w.gate_deg, threads=8)
        assert np.allclose(s[idx], so, rtol=0, atol=2e-6)
        assert bi == int(np.argmax(s)) and np.float32(bs) == s.max()


def test_three_objects_16k_hypotheses_each_full_size():
    """BASELINE.json configs[2] at its stated size: 3 objects, 16 384 hypotheses per object, C2-sized
    clouds (5000-point model, 50 000-point scene), weighted LCP on one GPU, then ICP refinement of the
    top 64 per object.  Over ALL hypotheses: size-independent properties (bounds, arg-max rule,
    invariance under a permutation of the batch and under splitting it, registered counts); the
    oracle on a sample of every object and on the best pose."""
    rng = np.random.default_rng(16384)
    for k in range(3):
        w = synth.make_workload(50000, 5000, 16384, config_id=210 + k)
        sc = LcpScorer()
        sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
        s, c, bi, bs = sc.score(w.T, PGP_MODE_WEIGHTED, w.gate_deg)
        sp, cp, bip, bsp = sc.score(w.T, PGP_MODE_PLAIN)
        n = len(w.T)
        assert n == 16384 and np.isfinite(s).all() and (s >= 0).all() and (s <= w.P_w.max() + 1e-6).all()
        assert (c >= 0).all() and (c <= cp).all() and (cp <= len(w.Q_xyz)).all()      # gated hits are hits
        assert np.array_equal(sp, cp.astype(np.float32) / np.float32(len(w.Q_xyz)))
        assert bi == int(np.argmax(s)) and np.float32(bs) == s[bi] and bip == int(np.argmax(sp))
        # a hypothesis' score does not depend on its place in the batch, nor on the batch it is in
        perm = rng.permutation(n)
        s2, c2, bi2, _ = sc.score(w.T[perm], PGP_MODE_WEIGHTED, w.gate_deg)
        assert np.array_equal(s2, s[perm]) and np.array_equal(c2, c[perm]) and perm[bi2] == bi
        half = n // 2 + 37
        sa = sc.score(w.T[:half], PGP_MODE_WEIGHTED, w.gate_deg)[0]
        sb = sc.score(w.T[half:], PGP_MODE_WEIGHTED, w.gate_deg)[0]
        # (bit for bit, except the few entries next to a batch maximum, which carry the reference-order sum)
        sab = np.concatenate([sa, sb])
        assert (sab != s).sum() <= 8 and np.allclose(sab, s, rtol=0, atol=2e-6)
        assert np.array_equal(LcpScorer.running_best(s), np.flatnonzero(s > np.concatenate([[0], np.maximum.accumulate(s)[:-1]])))
        # the oracle on a sample + the best pose
        orc = Oracle(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm)
        idx = np.unique(np.concatenate([np.arange(0, n, 499), [bi]]))
        so, _, _ = orc.score_batch(w.T[idx], w.delta, mode=1, gate_deg=w.gate_deg, threads=8)
        assert np.allclose(s[idx], so, rtol=0, atol=2e-6)
        ws, reg = orc.weighted_verify(w.T[bi], w.delta, w.gate_deg)
        assert np.array_equal(sc.registered(w.T[bi], PGP_MODE_WEIGHTED, w.gate_deg), reg) and c[bi] == len(reg)
        sop, _, _ = orc.score_batch(w.T[idx], w.delta, mode=0, threads=8)
        assert np.array_equal(sp[idx], sop)
        # ICP refinement of the top 64 (UCTState::performTrICP), refined poses re-scored
        top = np.argsort(-s, kind="stable")[:64]
        seg = w.P_xyz[w.P_w == 1.0]
        G = np.stack([_inv16(w.T[h]) for h in top])
        Gr, energy, iters = sc.icp_refine(seg, w.Q_xyz, G, trim=0.9, max_iterations=30)
        refined = np.stack([_inv16(g) for g in Gr])
        s_ref = sc.score(refined, PGP_MODE_PLAIN)[0]
        s_before = sp[top]
        assert s_ref.max() >= 0.97 * s_before.max() and np.median(s_ref) >= np.median(s_before)
        assert (iters >= 1).all() and np.isfinite(energy).all()
