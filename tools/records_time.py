#!/usr/bin/env python3
"""Cost of pgp_set_exact_records on the C2 batch (host-pointer scoring calls, 4096 hypotheses)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from physimglobalpose_amd import LcpScorer, synth, PGP_MODE_WEIGHTED
w = synth.make_workload(50000, 5000, 4096, config_id=2)
sc = LcpScorer(0)
sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
for on in (False, True, False, True):
    sc.set_exact_records(on)
    sc.score(w.T, PGP_MODE_WEIGHTED, w.gate_deg)
    t0 = time.perf_counter()
    for _ in range(50):
        s, _, bi, _ = sc.score(w.T, PGP_MODE_WEIGHTED, w.gate_deg)
    dt = (time.perf_counter() - t0) / 50
    print(f"exact_records={int(on)}: {dt*1e6:.1f} us per host-pointer call, {len(LcpScorer.running_best(s))} records, best {bi}")
