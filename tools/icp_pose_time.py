"""Time every pose spends inside the persistent ICP kernel (diagnostic library: make icpstamps LEVEL=4, copied over
physimglobalpose_amd/libpgp.so): the launch lasts as long as its slowest pose."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from physimglobalpose_amd import LcpScorer, synth
rng = np.random.default_rng(0)
M, _ = synth.make_model(rng, 5000); M = M.astype(np.float32)
R = synth._rot_axis_angle([0.2, 0.5, -0.4], 0.8); t = np.array([0.1, 0.0, 0.7])
S = (M[rng.choice(5000, 2500, replace=False)] @ R.T + t).astype(np.float32)
Tinv = np.linalg.inv(synth._se3(R, t))
sc = LcpScorer()
for n in (1, 64, 256):
    G = np.stack([synth.colmajor16(Tinv @ synth._se3(synth._random_rot(rng, np.deg2rad(5)), 0.005 * rng.standard_normal(3))) for _ in range(n)])
    for iters in (1, 10):
        T, e, it = sc.icp_refine(S, M, G, trim=0.9, max_iterations=iters)
        us = e.astype(np.float64) / 100.0
        print(f"poses {n:4d} max_iter {iters:2d}: iterations {it.min()}..{it.max()}  time in kernel per pose us: min {us.min():7.1f} mean {us.mean():7.1f} max {us.max():7.1f} (pose {int(us.argmax())})")
