"""Per-call timings of the '10% far outliers' ICP case of tools/icp_quick.py for a sweep of pose counts (every repetition
printed: a constant cost and a rare hiccup look different).  PGP_PKG_ROOT selects another checkout of the package (an
earlier round's build under tools/ab/), PGP_LIB another build of this round's library.
usage: python tools/icp_outlier_probe.py [reps] [pose counts ...]"""
import sys, os, time
ROOT = os.environ.get("PGP_PKG_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from physimglobalpose_amd import LcpScorer, synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
counts = [int(v) for v in sys.argv[2:]] or [64, 128, 192, 256, 320, 512, 1024]
rng = np.random.default_rng(0)
M, _ = synth.make_model(rng, 5000); M = M.astype(np.float32)
R = synth._rot_axis_angle([0.2, 0.5, -0.4], 0.8); t = np.array([0.1, 0.0, 0.7])
S0 = (M[rng.choice(5000, 2500, replace=False)] @ R.T + t).astype(np.float32)
S1 = S0.copy()
k = rng.choice(2500, 250, replace=False)
S1[k] += rng.uniform(-0.15, 0.15, (250, 3)).astype(np.float32)
Tinv = np.linalg.inv(synth._se3(R, t))
sc = LcpScorer()
for n in counts:
    G = np.stack([synth.colmajor16(Tinv @ synth._se3(synth._random_rot(rng, np.deg2rad(5)), 0.005 * rng.standard_normal(3))) for _ in range(n)])
    sc.icp_refine(S1, M, G, trim=0.9, max_iterations=10)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        Tr, e, it = sc.icp_refine(S1, M, G, trim=0.9, max_iterations=10)
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"outliers poses {n:5d}: " + " ".join(f"{v:7.3f}" for v in ts) + f" ms  (iterations {it.sum()}, max per pose {it.max()})", flush=True)
