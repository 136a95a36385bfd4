"""ICP timing: pose-iterations per second of pgp_icp_refine at the configs[2] shape (2500-point segment,
5000-point model), per search path: exact LDS index with one persistent workgroup per pose (default),
index with host-driven iterations, exhaustive host-driven scan (round-2 default) -- with and without 10 %
far outliers in the segment.  usage: python tools/icp_time.py [iterations]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from physimglobalpose_amd import LcpScorer, synth

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(0)
M, _ = synth.make_model(rng, 5000); M = M.astype(np.float32)
R = synth._rot_axis_angle([0.2, 0.5, -0.4], 0.8); t = np.array([0.1, 0.0, 0.7])
S0 = (M[rng.choice(5000, 2500, replace=False)] @ R.T + t).astype(np.float32)
S1 = S0.copy()
k = rng.choice(2500, 250, replace=False)
S1[k] += rng.uniform(-0.15, 0.15, (250, 3)).astype(np.float32)
Tinv = np.linalg.inv(synth._se3(R, t))
sc = LcpScorer()
PATHS = [("index persistent", {"PGP_ICP_NN": "index", "PGP_ICP_PERSIST": "1"}),
         ("index split", {"PGP_ICP_NN": "index", "PGP_ICP_PERSIST": "0"}),
         ("scan split", {"PGP_ICP_NN": "scan"})]
for label, S in (("clean segment", S0), ("10% far outliers", S1)):
    for n in (1, 8, 64, 256, 1024):
        G = np.stack([synth.colmajor16(Tinv @ synth._se3(synth._random_rot(rng, np.deg2rad(5)), 0.005 * rng.standard_normal(3))) for _ in range(n)])
        ref = None
        for name, env in PATHS:
            if n >= 1024 and name == "scan split":
                continue
            for kk in ("PGP_ICP_NN", "PGP_ICP_PERSIST"):
                os.environ.pop(kk, None)
            os.environ.update(env)
            sc.icp_refine(S, M, G, trim=0.9, max_iterations=iters)
            t0 = time.perf_counter()
            for _ in range(3): Tr, e, it = sc.icp_refine(S, M, G, trim=0.9, max_iterations=iters)
            dt = (time.perf_counter() - t0) / 3
            same = "" if ref is None else ("  identical" if np.array_equal(ref, Tr) else "  DIFFERS")
            if ref is None: ref = Tr
            print(f"{label:18s} poses {n:5d} {name:17s}: {dt*1e3:8.2f} ms/call, {it.sum()/dt:11.0f} pose-iters/s{same}", flush=True)
