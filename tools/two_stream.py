#!/usr/bin/env python3
"""Do consecutive scoring steps overlap when they alternate between two contexts on two streams
(the tail of one launch under the head of the next)?  Same C2 batch, 200 steps."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from physimglobalpose_amd import LcpScorer, synth, PGP_MODE_PLAIN, PGP_MODE_WEIGHTED  # noqa: E402

w = synth.make_workload(50000, 5000, 4096, config_id=2)
dT = torch.from_numpy(w.T).cuda()
for n_ctx in (1, 2, 3):
    scs, outs, streams = [], [], []
    for k in range(n_ctx):
        sc = LcpScorer(0)
        sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
        sc.reserve(4096)
        scs.append(sc)
        outs.append((torch.zeros(4096, device="cuda"), torch.zeros(4096, dtype=torch.int32, device="cuda"),
                     torch.zeros(2, dtype=torch.int32, device="cuda")))
        streams.append(torch.cuda.Stream())
    for mode, name in ((PGP_MODE_PLAIN, "plain"), (PGP_MODE_WEIGHTED, "weighted")):
        best = []
        for rep in range(5):
            for i in range(220):
                if i == 20:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                k = i % n_ctx
                scs[k].score_device(dT, *outs[k], mode=mode, stream=streams[k])
            torch.cuda.synchronize()
            best.append((time.perf_counter() - t0) / 200 * 1e6)
        print(f"{n_ctx} context(s) / stream(s)  {name:9s} step {min(best):6.1f} us  best index {int(outs[0][2][0])}")
