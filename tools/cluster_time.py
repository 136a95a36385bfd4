import sys, os, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from physimglobalpose_amd import LcpScorer, PGP_MODE_WEIGHTED, synth
w = synth.make_workload(50000, 5000, 16384, config_id=210)
sc = LcpScorer(); sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
s, _, _, bs = sc.score(w.T, PGP_MODE_WEIGHTED, w.gate_deg)
for frac in (0.5, 0.2, 0.0):
    sc.cluster_poses(w.T, s, bs, accept_fraction=frac)
    t0 = time.perf_counter()
    for _ in range(10):
        rep, asg = sc.cluster_poses(w.T, s, bs, accept_fraction=frac)
    dt = (time.perf_counter() - t0) / 10
    print(f"accept {frac}: {dt*1e3:.3f} ms, clusters {len(rep)}, kept {(asg >= 0).sum()}")
