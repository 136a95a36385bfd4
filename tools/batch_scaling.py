#!/usr/bin/env python3
"""Scoring step (score + finalize, resident inputs) against the batch size: 4096 .. 65536 hypotheses of the
C2 clouds on one GPU -- how much of a 4096-hypothesis step is fixed cost."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from physimglobalpose_amd import LcpScorer, synth, PGP_MODE_PLAIN, PGP_MODE_WEIGHTED

w = synth.make_workload(50000, 5000, 65536, config_id=2)
sc = LcpScorer(0)
sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
sc.reserve(65536)
for n in (1024, 4096, 16384, 65536):
    dT = torch.from_numpy(w.T[:n]).cuda()
    ds = torch.zeros(n, device="cuda")
    db = torch.zeros(2, dtype=torch.int32, device="cuda")
    for mode, name in ((PGP_MODE_PLAIN, "plain"), (PGP_MODE_WEIGHTED, "weighted")):
        best = []
        reps = max(10, 200 * 4096 // n)
        for _ in range(3):
            for _ in range(5):
                sc.score_device(dT, ds, None, db, mode=mode)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                sc.score_device(dT, ds, None, db, mode=mode)
            torch.cuda.synchronize()
            best.append((time.perf_counter() - t0) / reps)
        dt = min(best)
        print(f"{n:6d} hypotheses {name:8s}: {dt*1e6:8.1f} us per step, {n/dt/1e6:6.1f} M hyp/s, {dt*1e6*4096/n:6.1f} us per 4096", flush=True)
