#!/usr/bin/env python3
"""Sparse vs dense form of the scene index (csrc/grid_index.hip): build time, table size and scoring rate
on (a) the C2 tabletop scene, both forms forced, and (b) a 5 m room at delta = 5 mm, where the dense form
has to grow its cell.  Prints one line per case; profiles/r03_sparse_index.txt keeps the output."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from physimglobalpose_amd import LcpScorer, PGP_MODE_WEIGHTED, PGP_MODE_PLAIN, synth  # noqa: E402
from test_sparse_index_gpu import _room, _hypotheses  # noqa: E402


def rate(sc, T, mode, reps=20):
    d_T = torch.from_numpy(T).cuda()
    d_s = torch.empty(len(T), dtype=torch.float32, device="cuda")
    sc.reserve(len(T))
    for _ in range(3):
        sc.score_device(d_T, d_s, mode=mode)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        sc.score_device(d_T, d_s, mode=mode)
    torch.cuda.synchronize()
    return len(T) * reps / (time.perf_counter() - t0)


def case(name, P, Pn, Pw, Q, Qn, delta, T):
    for form in ("dense", "sparse"):
        os.environ["PGP_INDEX"] = form
        sc = LcpScorer()
        sc.init(P, Pn, Pw, Q, Qn, delta)
        sc.init(P, Pn, Pw, Q, Qn, delta)          # second build: buffers already allocated
        i = sc.index_info()
        rw, rp = rate(sc, T, PGP_MODE_WEIGHTED), rate(sc, T, PGP_MODE_PLAIN)
        print(f"{name:10s} {form:6s} cell {i['cell_size'] * 1e3:5.2f} mm  grid {i['grid_nx']}x{i['grid_ny']}x{i['grid_nz']}"
              f"  blocks {i['n_blocks']:>9d}  cand {i['n_candidates']:>8d}  index {i['bytes_index'] / 1e6:6.1f} MB"
              f"  build {i['build_ms']:6.2f} ms  weighted {rw / 1e6:5.1f} M hyp/s  plain {rp / 1e6:5.1f} M hyp/s", flush=True)
    os.environ.pop("PGP_INDEX")


w = synth.make_workload(50000, 5000, 65536, config_id=2)
case("C2", w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta, w.T)
rng = np.random.default_rng(40)
P, Pn, blobs = _room(rng, int(os.environ.get('ROOM_POINTS', '400000')))
Pw = rng.uniform(0.2, 1.0, len(P)).astype(np.float32)
obj = blobs[3].astype(np.float32)
Q = obj[rng.choice(len(obj), 2000, replace=False)]
Qn = synth._unit(rng.standard_normal(Q.shape)).astype(np.float32)
T = _hypotheses(rng, 16384, obj.mean(0), rot_deg=20.0, trans=0.05)
case("room 5 m", P, Pn, Pw, Q, Qn, 0.005, T)
