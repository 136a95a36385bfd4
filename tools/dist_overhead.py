#!/usr/bin/env python3
"""Where the per-step overhead of the multi-rank loop of bench.py comes from: one rank on the real
RCCL backend, the loop with pieces switched off (VARIANT: full, noargmax, nozero, noallreduce, none)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from physimglobalpose_amd import LcpScorer, synth, PGP_MODE_PLAIN  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29571")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
w = synth.make_workload(50000, 5000, 4096, config_id=2)
sc = LcpScorer(0)
sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
sc.reserve(4096)
dT = torch.from_numpy(w.T).cuda()
dc = torch.zeros(4096, dtype=torch.int32, device="cuda")
db = torch.zeros(2, dtype=torch.int32, device="cuda")
stream = torch.cuda.current_stream(dev)
post = torch.cuda.Stream(dev)
for variant in ("full", "noargmax", "nozero", "noallreduce", "none"):
    bufs = [torch.zeros(4096, device="cuda") for _ in range(2)]
    works = [None, None]
    ready = [torch.cuda.Event(), torch.cuda.Event()]
    k = 0

    def step():
        global k
        b = k % 2
        k += 1
        if variant != "none":
            with torch.cuda.stream(post):
                if works[b] is not None:
                    works[b].wait()
                    works[b] = None
                    if variant != "noargmax":
                        torch.argmax(bufs[b])
                if variant != "nozero":
                    bufs[b].zero_()
                ready[b].record(post)
            stream.wait_event(ready[b])
        sc.score_device(dT, bufs[b], dc, db, mode=PGP_MODE_PLAIN, stream=stream)
        if variant not in ("noallreduce", "none"):
            works[b] = dist.all_reduce(bufs[b], async_op=True)

    best = []
    for rep in range(4):
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            step()
        torch.cuda.synchronize()
        best.append((time.perf_counter() - t0) / 200 * 1e6)
    print(f"{variant:12s} step {min(best):6.1f} us")
dist.destroy_process_group()
