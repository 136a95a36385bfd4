"""Quick ICP timing of the default path only (tools/icp_time.py without the checker paths).
usage: python tools/icp_quick.py [iterations] [reps]"""
import sys, os, time, gc
gc.disable()   # a full collector pass (tens of ms, about every fifty calls) is not the library's time: tools/icp_hiccup_probe.py
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from physimglobalpose_amd import LcpScorer, synth

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rng = np.random.default_rng(0)
M, _ = synth.make_model(rng, 5000); M = M.astype(np.float32)
R = synth._rot_axis_angle([0.2, 0.5, -0.4], 0.8); t = np.array([0.1, 0.0, 0.7])
S0 = (M[rng.choice(5000, 2500, replace=False)] @ R.T + t).astype(np.float32)
S1 = S0.copy()
k = rng.choice(2500, 250, replace=False)
S1[k] += rng.uniform(-0.15, 0.15, (250, 3)).astype(np.float32)
Tinv = np.linalg.inv(synth._se3(R, t))
sc = LcpScorer()
for label, S, deg, tr in (("far start (5 deg, 5 mm)", S0, 5, 0.005), ("10% far outliers", S1, 5, 0.005), ("near start (0.3 deg, 1 mm)", S0, 0.3, 0.001)):
    for n in (1, 8, 64, 256, 1024):
        G = np.stack([synth.colmajor16(Tinv @ synth._se3(synth._random_rot(rng, np.deg2rad(deg)), tr * rng.standard_normal(3))) for _ in range(n)])
        sc.icp_refine(S, M, G, trim=0.9, max_iterations=iters)
        t0 = time.perf_counter()
        for _ in range(reps): Tr, e, it = sc.icp_refine(S, M, G, trim=0.9, max_iterations=iters)
        dt = (time.perf_counter() - t0) / reps
        print(f"{label:26s} poses {n:5d}: {dt*1e3:8.3f} ms/call, {it.sum()/dt/1e6:7.3f} M pose-iters/s  (iterations {it.sum()})", flush=True)
