#!/usr/bin/env python3
"""Is the scoring kernel limited by the memory system's locality?  Times (HIP events) the C2 batch as it
is, and batches made of ONE hypothesis repeated 4096 times (same work per hypothesis, perfect reuse of
every index line across the whole chip) for 32 sample hypotheses; the mean of the latter is what the
mixed batch would cost if locality were free."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from physimglobalpose_amd import LcpScorer, synth, PGP_MODE_PLAIN, PGP_MODE_WEIGHTED  # noqa: E402

w = synth.make_workload(50000, 5000, 4096, config_id=2)
sc = LcpScorer(0)
sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
sc.reserve(4096)
ds = torch.zeros(4096, device="cuda")
dc = torch.zeros(4096, dtype=torch.int32, device="cuda")
db = torch.zeros(2, dtype=torch.int32, device="cuda")
sc.set_kernel_timing(True)


def t(dT, mode, reps=20):
    for _ in range(5):
        sc.score_device(dT, ds, dc, db, mode=mode)
    torch.cuda.synchronize()
    sc.kernel_timing(reset=True)
    for _ in range(reps):
        sc.score_device(dT, ds, dc, db, mode=mode)
    torch.cuda.synchronize()
    n, ms = sc.kernel_timing(reset=True)
    return ms / n * 1e3


rng = np.random.default_rng(0)
sample = rng.choice(4096, 32, replace=False)
for mode, name in ((PGP_MODE_PLAIN, "plain"), (PGP_MODE_WEIGHTED, "weighted")):
    mixed = t(torch.from_numpy(w.T).cuda(), mode)
    rep = [t(torch.from_numpy(np.ascontiguousarray(np.repeat(w.T[h:h + 1], 4096, 0))).cuda(), mode, 5) for h in sample]
    print(f"{name:9s} mixed batch {mixed:6.1f} us | one hypothesis x 4096: mean {np.mean(rep):6.1f} us, "
          f"min {np.min(rep):6.1f}, max {np.max(rep):6.1f}")
