"""Looks for rare slow calls of pgp_icp_refine: N identical calls, the slow ones (> 3x the median) listed with their index.
usage: python tools/icp_hiccup_probe.py [calls] [poses] [gc: 0|1]"""
import sys, os, time, gc
ROOT = os.environ.get("PGP_PKG_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from physimglobalpose_amd import LcpScorer, synth

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 400
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
if len(sys.argv) > 3 and sys.argv[3] == "0":
    gc.disable()
rng = np.random.default_rng(0)
M, _ = synth.make_model(rng, 5000); M = M.astype(np.float32)
R = synth._rot_axis_angle([0.2, 0.5, -0.4], 0.8); t = np.array([0.1, 0.0, 0.7])
S = (M[rng.choice(5000, 2500, replace=False)] @ R.T + t).astype(np.float32)
Tinv = np.linalg.inv(synth._se3(R, t))
G = np.stack([synth.colmajor16(Tinv @ synth._se3(synth._random_rot(rng, np.deg2rad(5)), 0.005 * rng.standard_normal(3))) for _ in range(n)])
sc = LcpScorer()
ts = []
for _ in range(calls):
    t0 = time.perf_counter()
    sc.icp_refine(S, M, G, trim=0.9, max_iterations=10)
    ts.append((time.perf_counter() - t0) * 1e3)
ts = np.array(ts)
med = float(np.median(ts[1:]))
slow = [(i, round(float(v), 2)) for i, v in enumerate(ts) if i and v > 3 * med]
print(f"poses {n}: {calls} calls, median {med:.3f} ms, first {ts[0]:.2f} ms, slow calls (index, ms): {slow}", flush=True)
