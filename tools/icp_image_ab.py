#!/usr/bin/env python3
"""A/B of where the persistent ICP kernels keep the target's index image: in LDS (default: ~100 KB for a 5000-point model, so ONE
workgroup per compute unit = one wave per SIMD) or in memory / L2 (PGP_ICP_IMAGE=global: ~25 KB of LDS per workgroup, so several
workgroups per compute unit hide each other's latency -- but only when there are more poses than compute units).  The bench's
two regimes (from up to 6 cm off, from 1 mm off), 64 .. 4096 poses; identical results required.  usage: python tools/icp_image_ab.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from physimglobalpose_amd import LcpScorer, synth

rng = np.random.default_rng(0)
w = synth.make_workload(50000, 5000, 64, config_id=2)
seg = w.Q_xyz[rng.choice(len(w.Q_xyz), 2500, replace=False)]
R = synth._rot_axis_angle([0.2, 0.5, -0.4], 0.8)
S = (seg @ R.T + np.array([0.1, 0.0, 0.7])).astype(np.float32)
Tinv = np.linalg.inv(synth._se3(R, np.array([0.1, 0.0, 0.7])))
far = np.stack([synth.colmajor16(Tinv @ synth._se3(synth._random_rot(rng, np.deg2rad(5)), 0.005 * rng.standard_normal(3))) for _ in range(4096)])
near = np.stack([synth.colmajor16(Tinv @ synth._se3(synth._random_rot(rng, np.deg2rad(0.3)), 0.001 * rng.standard_normal(3))) for _ in range(4096)])
for regime, G in (("far", far), ("near", near)):
    for n in (64, 256, 512, 1024, 2048, 4096):
        res = {}
        for name, env in (("lds", None), ("global", "global")):
            os.environ.pop("PGP_ICP_IMAGE", None)
            if env:
                os.environ["PGP_ICP_IMAGE"] = env
            sc = LcpScorer(0)       # (the index is kept per context: a fresh one per form)
            sc.icp_refine(S, w.Q_xyz, G[:n], trim=0.9, max_iterations=10)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                out = sc.icp_refine(S, w.Q_xyz, G[:n], trim=0.9, max_iterations=10)
                ts.append(time.perf_counter() - t0)
            res[name] = (float(np.median(ts)), out)
            sc.close()
        same = all(np.array_equal(a, b) for a, b in zip(res["lds"][1], res["global"][1]))
        its = int(res["lds"][1][2].sum())
        print(f"{regime:4s} poses {n:5d}: LDS image {its / res['lds'][0] / 1e6:6.2f} M pose-it/s ({res['lds'][0]*1e3:7.3f} ms)   "
              f"global image {its / res['global'][0] / 1e6:6.2f} M ({res['global'][0]*1e3:7.3f} ms)   {'identical' if same else 'DIFFERS'}", flush=True)
