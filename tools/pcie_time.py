#!/usr/bin/env python3
"""Host-pointer vs device-pointer scoring call at C2 (PCIe-inclusive number for DESIGN.md)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from physimglobalpose_amd import LcpScorer, synth
w = synth.make_workload(50000, 5000, 4096, config_id=2)
sc = LcpScorer(0); sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta); sc.reserve(4096)
for _ in range(5): sc.score(w.T)
t0 = time.perf_counter()
for _ in range(50): sc.score(w.T)
host = (time.perf_counter() - t0) / 50
dT = torch.from_numpy(w.T).cuda(); ds = torch.zeros(4096, device="cuda"); db = torch.zeros(2, dtype=torch.int32, device="cuda")
for _ in range(5): sc.score_device(dT, ds, None, db)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): sc.score_device(dT, ds, None, db)
torch.cuda.synchronize(); dev = (time.perf_counter() - t0) / 50
print(f"host-pointer call {host*1e6:.1f} us ({4096/host/1e6:.1f} M hyp/s), device-pointer call {dev*1e6:.1f} us ({4096/dev/1e6:.1f} M hyp/s)")
