"""Which hypothesis of the bench workload differs from the oracle in weighted mode, and why (registered ids,
distances of the differing model point to its two candidate scene points)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from physimglobalpose_amd import LcpScorer, PGP_MODE_WEIGHTED, synth
from _checkers import Oracle
w = synth.make_workload(50000, 5000, 4096 * 8, config_id=2)
sc = LcpScorer(); sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
orc = Oracle(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm)
for exact in (False, True):
    sc.set_exact_ties(exact)
    if exact: sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    bad = []
    for b in range(8):
        T = w.T[b * 4096:(b + 1) * 4096]
        sw = sc.score(T, PGP_MODE_WEIGHTED, w.gate_deg)[0]
        swo = orc.score_batch(T, w.delta, mode=1, gate_deg=w.gate_deg, threads=16)[0]
        d = np.abs(sw.astype(np.float64) - swo)
        for i in np.flatnonzero(d > 2e-6): bad.append((b, int(i), float(d[i])))
    print(f"exact_ties={exact}: {len(bad)} hypotheses beyond 2e-6: {bad[:6]}")
    for b, i, dd in bad[:3]:
        T = w.T[b * 4096 + i]
        ws, reg = orc.weighted_verify(T, w.delta, w.gate_deg)
        mine = sc.registered(T, PGP_MODE_WEIGHTED, w.gate_deg)
        print(f"  batch {b} hyp {i}: oracle registers {len(reg)}, gpu {len(mine)}; only oracle {sorted(set(reg)-set(mine))}, only gpu {sorted(set(mine)-set(reg))}")
        only_o, only_g = sorted(set(reg) - set(mine)), sorted(set(mine) - set(reg))
        M = T.reshape(4, 4, order="F").astype(np.float32)
        Qt = (w.Q_xyz @ M[:3, :3].T + M[:3, 3]).astype(np.float32)
        for pid in only_o + only_g:
            p = w.P_xyz[pid]
            d2 = ((Qt - p) ** 2).sum(1)
            q = int(np.argmin(d2))
            dq = ((w.P_xyz - Qt[q]) ** 2).sum(1).astype(np.float32)
            order = np.argsort(dq)[:3]
            print(f"    scene point {pid} (weight {w.P_w[pid]:.3f}): model point {q}; its nearest scene points {order.tolist()} at d2 {dq[order].tolist()} (delta2 {w.delta**2:.3e})")
