#!/usr/bin/env python3
"""Wall time of pgp_cluster_poses (host pointers) on the C2 batch's own weighted scores."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, time
from physimglobalpose_amd import LcpScorer, synth, PGP_MODE_WEIGHTED
w = synth.make_workload(50000, 5000, 4096, config_id=2)
sc = LcpScorer(0); sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
sw,_,_,bs = sc.score(w.T, PGP_MODE_WEIGHTED, w.gate_deg)
for n in (100, 1024, 4096):
    T=w.T[:n]; s=sw[:n]+np.float32(1e-6)
    sc.cluster_poses(T,s,bs,accept_fraction=0.0)
    t0=time.perf_counter()
    for _ in range(20): rep,_=sc.cluster_poses(T,s,bs,accept_fraction=0.0)
    print(n, "poses:", round((time.perf_counter()-t0)/20*1e3,4), "ms per call,", len(rep), "clusters")
