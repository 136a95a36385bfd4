#!/usr/bin/env python3
"""What the fidelity switches cost at C2 (50 000 x 5 000 x 4096, weighted), one library per process:
    python tools/switches_cost.py            # the in-tree library
    PGP_LIB=tools/ab/libpgp_head.so python tools/switches_cost.py
default step / + pgp_set_exact_records / + pgp_set_exact_ties / both (what the drop-in runs for a segment with duplicates)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from physimglobalpose_amd import LcpScorer, PGP_MODE_WEIGHTED, synth  # noqa: E402

w = synth.make_workload(50000, 5000, 4096, config_id=2)
dT = torch.from_numpy(w.T).cuda()
ds = torch.zeros(4096, device="cuda")
name = os.environ.get("PGP_LIB", "in-tree")
for ties, rec in ((False, False), (False, True), (True, False), (True, True)):
    sc = LcpScorer()
    sc.set_exact_ties(ties)
    sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    sc.reserve(4096)
    sc.set_exact_records(rec)
    best = []
    for rep in range(5):
        for _ in range(20):
            sc.score_device(dT, ds, mode=PGP_MODE_WEIGHTED)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            sc.score_device(dT, ds, mode=PGP_MODE_WEIGHTED)
        torch.cuda.synchronize()
        best.append((time.perf_counter() - t0) / 200 * 1e6)
    print(f"{name}: exact ties {'on ' if ties else 'off'} exact records {'on ' if rec else 'off'}: weighted step {min(best):6.1f} us "
          f"(min of 5 x 200)  checksum {float(ds.double().sum()):.9f}", flush=True)
    del sc
