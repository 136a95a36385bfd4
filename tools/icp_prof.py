"""One ICP call per search path at 64 poses x 2500 x 5000 (10 iterations) for rocprofv3 --kernel-trace --stats."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from physimglobalpose_amd import LcpScorer, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
paths = sys.argv[2].split(",") if len(sys.argv) > 2 else ["persist", "split", "scan"]
rng = np.random.default_rng(0)
M, _ = synth.make_model(rng, 5000); M = M.astype(np.float32)
R = synth._rot_axis_angle([0.2, 0.5, -0.4], 0.8); t = np.array([0.1, 0.0, 0.7])
S = (M[rng.choice(5000, 2500, replace=False)] @ R.T + t).astype(np.float32)
Tinv = np.linalg.inv(synth._se3(R, t))
G = np.stack([synth.colmajor16(Tinv @ synth._se3(synth._random_rot(rng, np.deg2rad(5)), 0.005 * rng.standard_normal(3))) for _ in range(n)])
sc = LcpScorer()
ENV = {"persist": {"PGP_ICP_NN": "index", "PGP_ICP_PERSIST": "1"}, "split": {"PGP_ICP_NN": "index", "PGP_ICP_PERSIST": "0"},
       "scan": {"PGP_ICP_NN": "scan"}}
for p in paths:
    for kk in ("PGP_ICP_NN", "PGP_ICP_PERSIST"):
        os.environ.pop(kk, None)
    os.environ.update(ENV[p])
    for _ in range(3):
        sc.icp_refine(S, M, G, trim=0.9, max_iterations=int(os.environ.get("ICP_ITERS", "10")))
