#!/usr/bin/env python3
"""How well the lanes of a wave are balanced in the row search of the persistent ICP kernel (phase B): with the diagnostic
library of level 6 (make -C physimglobalpose_amd/csrc icpstamps LEVEL=6; PGP_LIB=tools/ab/libpgp_icpstamps.so) every
wave-pass adds its DEAREST lane's work (instruction units: 45 per row, 12 per point) and the mean over its 64 lanes; a wave
pays the dearest lane 64 times.  The bench's far-start problem (2500 x 5000, up to 6 cm / 5 degrees off)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from physimglobalpose_amd import LcpScorer, synth
rng = np.random.default_rng(0)
M, _ = synth.make_model(rng, 5000); M = M.astype(np.float32)
R = synth._rot_axis_angle([0.2, 0.5, -0.4], 0.8); t = np.array([0.1, 0.0, 0.7])
S = (M[rng.choice(5000, 2500, replace=False)] @ R.T + t).astype(np.float32)
Tinv = np.linalg.inv(synth._se3(R, t))
sc = LcpScorer()
for n in (64, 256):
    G = np.stack([synth.colmajor16(Tinv @ synth._se3(synth._random_rot(rng, np.deg2rad(5)), 0.005 * rng.standard_normal(3))) for _ in range(n)])
    for dp in (0, 1, 2, 3):
        os.environ["PGP_ICP_DBG_POSE"] = str(dp)
        T, e, it = sc.icp_refine(S, M, G, trim=0.9, max_iterations=10)
        dbg = e[8:16].astype(np.float64)
        if dbg[2] > 0:
            print(f"poses {n} pose {dp}: {int(it[dp])} iterations, {dbg[2]:.0f} wave-passes: dearest lane {dbg[0]/dbg[2]:7.0f} units per pass, "
                  f"mean lane {dbg[1]/dbg[2]:7.0f}: a balanced wave would take {dbg[1]/max(dbg[0],1):.2f} of the time")
