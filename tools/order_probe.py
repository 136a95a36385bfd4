#!/usr/bin/env python3
"""Does the ORDER of the hypotheses inside a batch matter to the scoring kernel?  Times the C2 batch as
given (random order) and sorted by where the pose puts the model's centroid (Morton code of the
translation, then rotation angle): hypotheses that share a workgroup then touch the same index cells."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from physimglobalpose_amd import LcpScorer, synth, PGP_MODE_PLAIN, PGP_MODE_WEIGHTED  # noqa: E402


def morton(t, lo, hi):
    c = np.clip((t - lo) / (hi - lo) * 1023, 0, 1023).astype(np.uint32)

    def spread(v):
        v = v & 1023
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    return spread(c[:, 0]) | (spread(c[:, 1]) << 1) | (spread(c[:, 2]) << 2)


w = synth.make_workload(50000, 5000, 4096, config_id=2)
sc = LcpScorer(0)
sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
sc.reserve(4096)
t = w.T[:, 12:15]
key = morton(t, t.min(0), t.max(0))
orders = {"as given": np.arange(4096), "sorted by translation (Morton)": np.argsort(key, kind="stable")}
ds = torch.zeros(4096, device="cuda")
dc = torch.zeros(4096, dtype=torch.int32, device="cuda")
db = torch.zeros(2, dtype=torch.int32, device="cuda")
sc.set_kernel_timing(True)
for name, o in orders.items():
    dT = torch.from_numpy(np.ascontiguousarray(w.T[o])).cuda()
    for mode, mn in ((PGP_MODE_PLAIN, "plain"), (PGP_MODE_WEIGHTED, "weighted")):
        for _ in range(10):
            sc.score_device(dT, ds, dc, db, mode=mode)
        torch.cuda.synchronize()
        sc.kernel_timing(reset=True)
        for _ in range(50):
            sc.score_device(dT, ds, dc, db, mode=mode)
        torch.cuda.synchronize()
        n, ms = sc.kernel_timing(reset=True)
        print(f"{name:32s} {mn:9s} kernel {ms / n * 1e3:7.1f} us")
