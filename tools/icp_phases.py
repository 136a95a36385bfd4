"""Where one ICP iteration of the persistent indexed kernel spends its time: run with the diagnostic
library (make -C physimglobalpose_amd/csrc icpstamps; cp tools/ab/libpgp_icpstamps.so physimglobalpose_amd/libpgp.so).
Thread 0 of ONE pose sums s_memrealtime deltas per phase (100 MHz ticks -> us); poses 16.. report their whole time
in the kernel, and the slowest of them is timed in a second run (a launch lasts as long as its slowest pose)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from physimglobalpose_amd import LcpScorer, synth
rng = np.random.default_rng(0)
M, _ = synth.make_model(rng, 5000); M = M.astype(np.float32)
R = synth._rot_axis_angle([0.2, 0.5, -0.4], 0.8); t = np.array([0.1, 0.0, 0.7])
S = (M[rng.choice(5000, 2500, replace=False)] @ R.T + t).astype(np.float32)
Tinv = np.linalg.inv(synth._se3(R, t))
sc = LcpScorer()


def report(n, G, iters, dp):
    os.environ["PGP_ICP_DBG_POSE"] = str(dp)
    T, e, it = sc.icp_refine(S, M, G, trim=0.9, max_iterations=iters)
    ni = it[dp]
    ticks = e[1:6].astype(np.float64)
    us = ticks / 100.0 / ni          # s_memrealtime: 100 MHz
    dbg = e[8:16].astype(np.float64); nq = max(dbg[4], 1)
    if dbg[4] > 0:
        print(f"    per query: rows {dbg[0]/nq:7.1f}  live rows {dbg[1]/nq:7.1f}  points {dbg[2]/nq:7.1f}  lanes {dbg[3]/nq:5.2f}   (queries {nq/ni:.0f} per iteration)")
    if os.environ.get("ICP_SUMS"):   # level-5 library: the parts of the sums phase
        print(f"    sums: tie ranking {dbg[0]/100/ni:5.1f}  accumulate {dbg[1]/100/ni:5.1f}  wave sums {dbg[2]/100/ni:5.1f}  barriers + final {dbg[3]/100/ni:5.1f} us per iteration")
    if os.environ.get("ICP_WAVES"):
        print(f"    search loop per wave (sum over iterations, us): mean {dbg[0]/100/16:.1f}  wave0 {dbg[2]/100:.1f}  last wave {dbg[3]/100:.1f}  (max single {dbg[1]/100:.1f})")
    print(f"    queries left to the row search per iteration: {dbg[3]/ni:.0f}")
    print(f"    nn split: bounds {dbg[5]/100/ni:6.1f}  sort {dbg[6]/100/ni:6.1f}  search {dbg[7]/100/ni:6.1f} us per iteration")
    print(f"poses {n:4d} pose {dp:3d} iterations {ni:3d}: per iteration  nn {us[0]:7.1f}  select {us[1]:7.1f}  sums+reduce {us[2]:7.1f}  solve {us[3]:7.1f}  stop rules {us[4]:7.1f}  us  (total {us.sum():7.1f})")
    whole = e[16:].astype(np.float64) / 100.0
    if len(whole):
        print(f"    whole time in the kernel of poses 16..: min {whole.min():.0f}  mean {whole.mean():.0f}  max {whole.max():.0f} us (pose {16 + int(whole.argmax())})")
        return 16 + int(whole.argmax())
    return dp


for n in (16, 64, 256):
    G = np.stack([synth.colmajor16(Tinv @ synth._se3(synth._random_rot(rng, np.deg2rad(5)), 0.005 * rng.standard_normal(3))) for _ in range(n)])
    for iters in (1, 10, 30):
        slow = report(n, G, iters, 0)
        if slow != 0:
            report(n, G, iters, slow)
