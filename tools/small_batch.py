#!/usr/bin/env python3
"""Small-batch scoring steps (score + finalize, resident inputs) at the C2 clouds: us per step against the
hypotheses-per-workgroup knob (PGP_HPB, read at pgp_create) -- what a 1024-hypothesis call pays beyond its share
of a large batch.  usage: python tools/small_batch.py [hpb ...]   (0 = the library's own choice)"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from physimglobalpose_amd import LcpScorer, synth, PGP_MODE_PLAIN, PGP_MODE_WEIGHTED

w = synth.make_workload(50000, 5000, 8192, config_id=2)
hpbs = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3, 4, 6, 8]
rng = np.random.default_rng(0)
perm = rng.permutation(8192)      # a representative mix at every size
for hpb in hpbs:
    if hpb:
        os.environ["PGP_HPB"] = str(hpb)
    else:
        os.environ.pop("PGP_HPB", None)
    sc = LcpScorer(0)
    sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    sc.reserve(8192)
    row = []
    for n in (8, 256, 512, 1024, 2048, 4096):
        dT = torch.from_numpy(w.T[perm[:n]]).cuda()
        ds = torch.zeros(n, device="cuda")
        db = torch.zeros(2, dtype=torch.int32, device="cuda")
        for mode, name in ((PGP_MODE_WEIGHTED, "w"),):
            best = []
            reps = 400
            for _ in range(3):
                for _ in range(20):
                    sc.score_device(dT, ds, None, db, mode=mode)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    sc.score_device(dT, ds, None, db, mode=mode)
                torch.cuda.synchronize()
                best.append((time.perf_counter() - t0) / reps)
            dt = min(best)
            row.append(f"{n}: {dt*1e6:6.1f} us ({n/dt/1e6:5.1f} M/s)")
    print(f"hpb {hpb or 'auto':>4}  weighted  " + "  ".join(row), flush=True)
    del sc
