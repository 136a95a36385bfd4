"""extract_pairs / find_congruent / rigid fits timed as bench.py's rows do (A/B through PGP_LIB)."""
import sys, os, time, gc
ROOT = os.environ.get("PGP_PKG_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from physimglobalpose_amd import LcpScorer, synth
gc.disable()
def timed(fn, reps=20):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    return (time.perf_counter() - t0) / reps, r
rng = np.random.default_rng(0)
w = synth.make_workload(50000, 5000, 64, config_id=2)
sc = LcpScorer(0)
sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
w2 = synth.make_workload(4000, 2000, 4, config_id=3, n_search=1000)
sc.set_search_model(w2.Qs_xyz)
T = w2.T_gt.reshape(4, 4).T
ids = rng.choice(1000, 4, replace=False)
base = (w2.Qs_xyz[ids] @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
d1 = float(np.linalg.norm(base[0] - base[1])); d6 = float(np.linalg.norm(base[2] - base[3]))
for rep in range(3):
    dt1, p1 = timed(lambda: sc.extract_pairs(d1, w.delta, cap=1 << 20))
    p6 = sc.extract_pairs(d6, w.delta, cap=1 << 20)
    dt2, q = timed(lambda: sc.find_congruent(base, 0.4, 0.6, w.delta, p1, p6, cap=1 << 20))
    b = rng.integers(0, len(w.P_xyz), (10000, 4)).astype(np.int32)
    qd = rng.integers(0, 1000, (10000, 4)).astype(np.int32)
    dt3, _ = timed(lambda: sc.rigid_from_congruent(b, qd, w.centroid_P, w.centroid_Q))
    print(f"extract {dt1*1e3:.4f} ms ({len(p1)} pairs)  find {dt2*1e3:.4f} ms ({len(q)} quads)  rigid {dt3*1e3:.4f} ms", flush=True)
