#!/usr/bin/env python3
"""The two ICP cases of bench.py side by side, for A/B of launch knobs (PGP_ICP_WGS, PGP_ICP_SOLO_TICKS, ...):
  far  = the `icp` row: 64 / 256 poses x 2500 x 5000, guesses up to 6 cm off, 10 iterations;
  near = the ICP of `config2_object`: the 64 best of 16 384 scored hypotheses, refined from their own poses (30 iterations).
usage: python tools/icp_regimes.py [label]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from physimglobalpose_amd import LcpScorer, PGP_MODE_WEIGHTED, synth  # noqa: E402

label = sys.argv[1] if len(sys.argv) > 1 else ""


def timed(fn, reps=5):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    return (time.perf_counter() - t0) / reps, r


rng = np.random.default_rng(0)
w = synth.make_workload(50000, 5000, 16384, config_id=210)
sc = LcpScorer()
sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
seg = w.Q_xyz[rng.choice(len(w.Q_xyz), 2500, replace=False)]
R = synth._rot_axis_angle([0.2, 0.5, -0.4], 0.8)
S = (seg @ R.T + np.array([0.1, 0.0, 0.7])).astype(np.float32)
Tinv = np.linalg.inv(synth._se3(R, np.array([0.1, 0.0, 0.7])))
G_all = np.stack([synth.colmajor16(Tinv @ synth._se3(synth._random_rot(rng, np.deg2rad(5)), 0.005 * rng.standard_normal(3)))
                  for _ in range(256)])
row = []
for n_p in (1, 8, 64, 128, 256):
    dt, (_, _, its) = timed(lambda: sc.icp_refine(S, w.Q_xyz, G_all[:n_p], trim=0.9, max_iterations=10))
    row.append(f"far {n_p}: {dt * 1e3:.3f} ms {its.sum() / dt / 1e3:.0f} k/s")
s, _, _, _ = sc.score(w.T, PGP_MODE_WEIGHTED, w.gate_deg)
top = np.argsort(-s, kind="stable")[:64]
segP = np.ascontiguousarray(w.P_xyz[w.P_w == 1.0])
G = np.stack([synth.colmajor16(np.linalg.inv(np.asarray(w.T[h], np.float64).reshape(4, 4).T)) for h in top])
for n_p in (8, 64):
    dt, (_, _, its) = timed(lambda: sc.icp_refine(segP, w.Q_xyz, G[:n_p], trim=0.9, max_iterations=30))
    row.append(f"near {n_p}: {dt * 1e3:.3f} ms {its.sum() / dt / 1e3:.0f} k/s")
print(f"{label:24s} " + " | ".join(row), flush=True)
