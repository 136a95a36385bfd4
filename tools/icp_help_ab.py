"""Helping (finished workgroups take search passes of running poses, csrc/icp.hip HelpPub) against the plain
one-workgroup-per-pose launch: same bits, time per call.  usage: python tools/icp_help_ab.py"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from physimglobalpose_amd import LcpScorer, synth
rng = np.random.default_rng(0)
M, _ = synth.make_model(rng, 5000); M = M.astype(np.float32)
R = synth._rot_axis_angle([0.2, 0.5, -0.4], 0.8); t = np.array([0.1, 0.0, 0.7])
S0 = (M[rng.choice(5000, 2500, replace=False)] @ R.T + t).astype(np.float32)
S1 = S0.copy(); k = rng.choice(2500, 250, replace=False); S1[k] += rng.uniform(-0.15, 0.15, (250, 3)).astype(np.float32)
Tinv = np.linalg.inv(synth._se3(R, t))
sc = LcpScorer()
for label, S, deg, tr in (("far start", S0, 5, 0.005), ("10% outliers", S1, 5, 0.005), ("near start", S0, 0.3, 0.001)):
    for n in (130, 192, 256):
        G = np.stack([synth.colmajor16(Tinv @ synth._se3(synth._random_rot(rng, np.deg2rad(deg)), tr * rng.standard_normal(3))) for _ in range(n)])
        out = {}
        for help_ in ("0", "1"):
            os.environ["PGP_ICP_HELP"] = help_
            sc.icp_refine(S, M, G, trim=0.9, max_iterations=10)
            t0 = time.perf_counter()
            for _ in range(5): r = sc.icp_refine(S, M, G, trim=0.9, max_iterations=10)
            out[help_] = ((time.perf_counter() - t0) / 5, r)
        same = all(np.array_equal(a, b) for a, b in zip(out["0"][1], out["1"][1]))
        its = out["0"][1][2].sum()
        print(f"{label:13s} poses {n:4d}: plain {out['0'][0]*1e3:7.3f} ms ({its/out['0'][0]/1e6:5.2f} M/s)   helping {out['1'][0]*1e3:7.3f} ms ({its/out['1'][0]/1e6:5.2f} M/s)   same bits: {same}", flush=True)
