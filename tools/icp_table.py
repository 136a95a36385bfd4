#!/usr/bin/env python3
"""The reference's table alignment (SceneCfg.cpp:101,135-141): a 30 000-point scene against a 100 000-point table, one pose,
max correspondence distance 1 cm -- wall time of the call and of its kernels (run under rocprofv3 --kernel-trace --stats for
the per-kernel split)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from physimglobalpose_amd import LcpScorer, synth
rng = np.random.default_rng(12)
top = np.c_[rng.uniform(-0.6, 0.6, 90000), rng.uniform(-0.4, 0.4, 90000), 0.0005 * rng.standard_normal(90000)]
rim = np.c_[rng.uniform(-0.6, 0.6, 10000), np.where(rng.random(10000) < 0.5, -0.4, 0.4), rng.uniform(-0.05, 0.0, 10000)]
tgt = np.concatenate([top, rim]).astype(np.float32)
R = synth._random_rot(rng, np.deg2rad(1.0))
pick = rng.choice(len(tgt), 30000, replace=False)
src = (tgt[pick] @ R.T + np.array([0.004, -0.003, 0.002]) + 0.0008 * rng.standard_normal((30000, 3))).astype(np.float32)
src[:300] += rng.uniform(-0.2, 0.2, (300, 3)).astype(np.float32)
G0 = synth.colmajor16(np.eye(4))[None]
sc = LcpScorer()
kw = dict(max_iterations=int(os.environ.get("ICP_TABLE_ITERS", "30")), max_corr_dist=0.01, energy_ratio=0.0, transformation_epsilon=1e-9, absolute_mse=1e-12)
sc.icp_refine_ex(src, tgt, G0, **kw)
t0 = time.perf_counter()
n = 5
for _ in range(n):
    T, E, it = sc.icp_refine_ex(src, tgt, G0, **kw)
dt = (time.perf_counter() - t0) / n
print(f"table alignment 30 000 x 100 000, cap 1 cm: {dt * 1e3:.2f} ms per call, {int(it[0])} iterations ({dt * 1e6 / max(int(it[0]), 1):.0f} us per iteration), rms {float(np.sqrt(E[0])) * 1e3:.2f} mm")
