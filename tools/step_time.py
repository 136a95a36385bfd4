#!/usr/bin/env python3
"""Wall time per asynchronous scoring step (score + finalize launches, no host sync inside the
loop) -- what bench.py's ms_per_step measures, without its event records."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from physimglobalpose_amd import LcpScorer, synth, PGP_MODE_PLAIN, PGP_MODE_WEIGHTED  # noqa: E402

w = synth.make_workload(50000, 5000, 4096, config_id=2)
sc = LcpScorer(0)
sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
sc.reserve(4096)
dT = torch.from_numpy(w.T).cuda()
ds = torch.zeros(4096, device="cuda")
dc = torch.zeros(4096, dtype=torch.int32, device="cuda")
db = torch.zeros(2, dtype=torch.int32, device="cuda")
for mode, name in ((PGP_MODE_PLAIN, "plain"), (PGP_MODE_WEIGHTED, "weighted")):
    best = []
    for rep in range(5):
        for _ in range(20):
            sc.score_device(dT, ds, dc, db, mode=mode)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            sc.score_device(dT, ds, dc, db, mode=mode)
        torch.cuda.synchronize()
        best.append((time.perf_counter() - t0) / 200 * 1e6)
    print(f"{name}: step {min(best):.1f} us (min of 5 x 200), best index {int(db[0])}")

# the same steps replayed from a HIP graph (one captured scoring call per replay)
for mode, name in ((PGP_MODE_PLAIN, "plain"), (PGP_MODE_WEIGHTED, "weighted")):
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        sc.score_device(dT, ds, dc, db, mode=mode, stream=side)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            sc.score_device(dT, ds, dc, db, mode=mode, stream=side)
    torch.cuda.current_stream().wait_stream(side)
    best = []
    for rep in range(5):
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            g.replay()
        torch.cuda.synchronize()
        best.append((time.perf_counter() - t0) / 200 * 1e6)
    print(f"{name}: graph replay step {min(best):.1f} us (min of 5 x 200), best index {int(db[0])}")
