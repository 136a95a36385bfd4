"""Device time line of ONE call of a host-pointer entry point (the 15th of 20), for the side rows of bench.py: which operations a
call queues, their durations and the gaps between them.  Run under rocprofv3 by tools/call_timeline.sh.
usage: python tools/call_timeline.py voxel|mls|backproject|cluster"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from physimglobalpose_amd import LcpScorer, synth
what = sys.argv[1]
rng = np.random.default_rng(0)
sc = LcpScorer(0)
if what in ("voxel", "mls"):
    w = synth.make_workload(20000, 3000, 4, config_id=5)
    seg = (w.P_xyz + 0.0007 * rng.standard_normal(w.P_xyz.shape)).astype(np.float32)
    vox = sc.voxel_grid(seg, 0.005)
    fn = (lambda: sc.voxel_grid(seg, 0.005)) if what == "voxel" else (lambda: sc.mls_normals(vox if not isinstance(vox, tuple) else vox[0], 0.01))
elif what == "backproject":
    yy, xx = np.mgrid[0:480, 0:640]
    depth = (0.8 + 0.1 * np.sin(xx / 50.0) + 0.05 * np.cos(yy / 40.0)).astype(np.float32)
    K = np.array([[600, 0, 320], [0, 600, 240], [0, 0, 1]], np.float32)
    mask = ((xx - 320) ** 2 + (yy - 240) ** 2 < 120 ** 2).astype(np.uint8)
    fn = lambda: sc.backproject_depth(depth, K, mask)
else:
    raise SystemExit("unknown row")
for k in range(20):
    if k == 14:
        time.sleep(0.02)      # an idle gap in front of the call that is looked at
    t0 = time.perf_counter(); fn(); dt = time.perf_counter() - t0
    if k == 14:
        print(f"CALL_MS {dt * 1e3:.3f}")
        time.sleep(0.02)
