#!/usr/bin/env python3
"""Does the ORDER of the model points matter to the scoring step?  The same C2 workload with the model as generated and
with its points sorted along a Morton curve (spatially compact groups of 64 = one wave): same hypotheses, same counts, the
float scores equal up to summation order.  Experiment of round 5."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from physimglobalpose_amd import LcpScorer, synth, PGP_MODE_PLAIN, PGP_MODE_WEIGHTED  # noqa: E402


def morton_order(xyz, bits=10):
    lo, hi = xyz.min(0), xyz.max(0)
    q = np.minimum(((xyz - lo) / np.maximum(hi - lo, 1e-12) * (1 << bits)).astype(np.int64), (1 << bits) - 1)
    code = np.zeros(len(xyz), np.int64)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return np.argsort(code, kind="stable")


def step_us(sc, dT, mode):
    ds = torch.zeros(dT.shape[0], device="cuda")
    dc = torch.zeros(dT.shape[0], dtype=torch.int32, device="cuda")
    db = torch.zeros(2, dtype=torch.int32, device="cuda")
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
    return min(best), ds.cpu().numpy(), dc.cpu().numpy(), int(db[0])


n_h = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
w = synth.make_workload(50000, 5000, n_h, config_id=2)
dT = torch.from_numpy(w.T).cuda()
orders = {"as generated": np.arange(len(w.Q_xyz)), "morton": morton_order(w.Q_xyz), "shuffled": np.random.default_rng(1).permutation(len(w.Q_xyz))}
ref = None
for _ in range(2):
    for name, perm in orders.items():
        sc = LcpScorer(0)
        sc.init(w.P_xyz, w.P_nrm, w.P_w, np.ascontiguousarray(w.Q_xyz[perm]), np.ascontiguousarray(w.Q_nrm[perm]), w.delta)
        sc.reserve(n_h)
        out = []
        for mode, mname in ((PGP_MODE_PLAIN, "plain"), (PGP_MODE_WEIGHTED, "weighted")):
            us, s, c, b = step_us(sc, dT, mode)
            if ref is None:
                ref = {}
            key = mname
            if key not in ref:
                ref[key] = (s, c, b)
            same_c = bool(np.array_equal(c, ref[key][1]))
            ds_max = float(np.abs(s - ref[key][0]).max() / max(np.abs(ref[key][0]).max(), 1e-30))
            out.append(f"{mname} {us:6.1f} us (counts equal {same_c}, scores within {ds_max:.1e} rel, best {b})")
        print(f"{name:13s}: " + "; ".join(out), flush=True)
        sc.close()
