"""One ICP configuration of tools/icp_quick.py, a few calls (for counter passes): python tools/icp_one.py [poses] [deg] [calls]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from physimglobalpose_amd import LcpScorer, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
deg = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rng = np.random.default_rng(0)
M, _ = synth.make_model(rng, 5000); M = M.astype(np.float32)
R = synth._rot_axis_angle([0.2, 0.5, -0.4], 0.8); t = np.array([0.1, 0.0, 0.7])
S = (M[rng.choice(5000, 2500, replace=False)] @ R.T + t).astype(np.float32)
Tinv = np.linalg.inv(synth._se3(R, t))
G = np.stack([synth.colmajor16(Tinv @ synth._se3(synth._random_rot(rng, np.deg2rad(deg)), 0.001 * deg * rng.standard_normal(3))) for _ in range(n)])
sc = LcpScorer()
for _ in range(calls):
    T, e, it = sc.icp_refine(S, M, G, trim=0.9, max_iterations=10)
print("poses", n, "iterations", int(it.sum()))
