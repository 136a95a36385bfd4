import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from physimglobalpose_amd import LcpScorer, synth
rng = np.random.default_rng(0)
M, N = synth.make_model(rng, 2000)
P = (M + np.array([0.1, 0.0, 0.7])).astype(np.float32)
sc = LcpScorer()
for k in range(3): sc.set_scene(P, N.astype(np.float32), None, 0.005)
t0 = time.perf_counter()
for k in range(50): sc.set_scene(P, N.astype(np.float32), None, 0.005)
print(f"PGP_BUILD_SMALL={os.environ.get('PGP_BUILD_SMALL','1')}: set_scene of a compact 2000-point object: {(time.perf_counter()-t0)/50*1e3:.3f} ms; index {sc.index_info()['n_cells']} cells, build on device {sc.index_info()['build_ms']:.3f} ms")
