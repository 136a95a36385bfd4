#!/usr/bin/env python3
"""Per-call latency of the synchronous scoring call (SURVEY 8(d)'s metric) on the C2 workload: pgp_score_lcp (host pointers) and
pgp_score_lcp_device + sync, median / p99 of 400 calls each at 4096 and 3000 hypotheses; with PGP_CALL_PHASES=1 the library
prints where the host-pointer call's time goes.  Usage: python tools/per_call_probe.py [n_calls]"""
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from physimglobalpose_amd import LcpScorer, PGP_MODE_WEIGHTED, PGP_MODE_PLAIN, synth  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 400
w = synth.make_workload(50000, 5000, 4096 * 8, config_id=2)
sc = LcpScorer(0)
sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
sc.reserve(4096)
Th = w.T.reshape(8, 4096, 16)
Td = [torch.from_numpy(Th[b]).cuda() for b in range(8)]
ds = torch.zeros(4096, device="cuda")
dc = torch.zeros(4096, dtype=torch.int32, device="cuda")
db = torch.zeros(2, dtype=torch.int32, device="cuda")
gc.disable()
for mode, name in ((PGP_MODE_WEIGHTED, "weighted"), (PGP_MODE_PLAIN, "plain")):
    for n in (4096, 3000, 1024, 256):
        hs, dv = [], []
        for k in range(calls + 50):
            T = Th[k % 8][:n]
            t0 = time.perf_counter()
            sc.score(T, mode, w.gate_deg)
            if k >= 50:
                hs.append(time.perf_counter() - t0)
        for k in range(calls + 50):
            t0 = time.perf_counter()
            sc.score_device(Td[k % 8][:n], ds[:n], dc[:n], db, mode=mode, gate_deg=w.gate_deg)
            torch.cuda.synchronize()
            if k >= 50:
                dv.append(time.perf_counter() - t0)
        # the stream figure for the same size
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(calls):
            sc.score_device(Td[k % 8][:n], ds[:n], dc[:n], db, mode=mode, gate_deg=w.gate_deg)
        torch.cuda.synchronize()
        st = (time.perf_counter() - t0) / calls
        hs, dv = np.sort(hs) * 1e3, np.sort(dv) * 1e3
        print(f"{name:8s} n_h {n:5d}: host pointers median {hs[len(hs)//2]:.4f} p99 {hs[int(0.99*len(hs))]:.4f} ms | "
              f"device + sync median {dv[len(dv)//2]:.4f} p99 {dv[int(0.99*len(dv))]:.4f} ms | stream {st*1e3:.4f} ms", flush=True)
