#!/usr/bin/env python3
"""Uncapped ICP against a target beyond the exact index's 65 535 points (a 100 000-point table, 30 000-point scene):
the open grid + scan of the unsettled queries (default) against the exhaustive scan alone (nn_search = 1), from close by
and from 8 cm off.  Same transforms either way (tests/test_icp_variants_gpu.py); this prints the times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from physimglobalpose_amd import LcpScorer, synth
rng = np.random.default_rng(21)
top = np.c_[rng.uniform(-0.6, 0.6, 90000), rng.uniform(-0.4, 0.4, 90000), 0.0005 * rng.standard_normal(90000)]
rim = np.c_[rng.uniform(-0.6, 0.6, 10000), np.where(rng.random(10000) < 0.5, -0.4, 0.4), rng.uniform(-0.05, 0.0, 10000)]
tgt = np.concatenate([top, rim]).astype(np.float32)
R = synth._random_rot(rng, np.deg2rad(1.0))
pick = rng.choice(len(tgt), 30000, replace=False)
sc = LcpScorer()
for name, off in (("close (4 mm off)", [0.004, -0.003, 0.002]), ("far (8 cm off)", [0.05, -0.04, 0.05])):
    src = (tgt[pick] @ R.T + np.array(off) + 0.0008 * rng.standard_normal((30000, 3))).astype(np.float32)
    G0 = synth.colmajor16(np.eye(4))[None]
    kw = dict(max_iterations=10, trim_fraction=0.9, energy_ratio=0.0, transformation_epsilon=1e-9, absolute_mse=1e-12)
    out = {}
    for nn in (1, 0):
        sc.icp_refine_ex(src, tgt, G0, nn_search=nn, **kw)
        t0 = time.perf_counter()
        for _ in range(3):
            T, E, it = sc.icp_refine_ex(src, tgt, G0, nn_search=nn, **kw)
        out[nn] = ((time.perf_counter() - t0) / 3 * 1e3, T, int(it[0]))
    print(f"{name}: exhaustive scan {out[1][0]:.2f} ms, open grid + scan of the rest {out[0][0]:.2f} ms "
          f"({out[0][2]} iterations, same transform: {np.array_equal(out[0][1], out[1][1])})", flush=True)
