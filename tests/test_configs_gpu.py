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
