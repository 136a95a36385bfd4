#!/usr/bin/env python3
"""The PCL form of the reference's ICP calls (greedy_bfs/State.cpp:139-142: 50 iterations, max correspondence distance 1 cm,
transformation epsilon 1e-8, absolute MSE 1e-12) on 2500 x 5000 points -- the persistent kernels WITH the extra stop rules."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from physimglobalpose_amd import LcpScorer, synth
rng = np.random.default_rng(0)
M, _ = synth.make_model(rng, 5000); M = M.astype(np.float32)
R = synth._rot_axis_angle([0.2, 0.5, -0.4], 0.8); t = np.array([0.1, 0.0, 0.7])
S = (M[rng.choice(5000, 2500, replace=False)] @ R.T + t).astype(np.float32)
Tinv = np.linalg.inv(synth._se3(R, t))
sc = LcpScorer()
kw = dict(max_iterations=50, max_corr_dist=0.01, energy_ratio=0.0, transformation_epsilon=1e-8, absolute_mse=1e-12)
for n in (1, 64, 256):
    G = np.stack([synth.colmajor16(Tinv @ synth._se3(synth._random_rot(rng, np.deg2rad(1.0)), 0.002 * rng.standard_normal(3))) for _ in range(n)])
    sc.icp_refine_ex(S, M, G, **kw)
    t0 = time.perf_counter()
    for _ in range(5):
        T, E, it = sc.icp_refine_ex(S, M, G, **kw)
    dt = (time.perf_counter() - t0) / 5
    print(f"{os.environ.get('PGP_LIB', 'in-tree'):30s} PCL form, {n:4d} poses: {dt * 1e3:7.3f} ms per call, {int(it.sum())} iterations in all, "
          f"checksum {float(np.abs(T).sum()):.9f}", flush=True)
