#!/usr/bin/env python3
"""VERDICT r4 task 8: could the exact-tie detector of the weighted scoring kernel (a second LDS atomic per IN-RANGE candidate,
+20 % of the step) be gated per owner -- taken only by queries whose candidate run holds two or more in-range candidates?
The gate can only help if most matched queries have exactly ONE scene point within delta.  This counts, on the bench's own
C2 workload (50 000-point scene, 5 000-point model, delta 5 mm, the 8 x 4096 hypotheses of bench.py), how many scene points
lie within delta of every transformed model point (CPU, scipy cKDTree, float64 -- a frequency, not a parity statement).
Writes profiles/r05_ab/tie_gate_frequency.json."""
import json
import os
import sys

import numpy as np
from scipy.spatial import cKDTree

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from physimglobalpose_amd import synth  # noqa: E402

w = synth.make_workload(50000, 5000, 4096 * 8, config_id=2)
tree = cKDTree(w.P_xyz.astype(np.float64))
rng = np.random.default_rng(0)
sample = rng.choice(len(w.T), 512, replace=False)          # 512 of the 32 768 hypotheses
Q = w.Q_xyz.astype(np.float64)
hist = np.zeros(16, np.int64)
for h in sample:
    M = w.T[h].reshape(4, 4, order="F").astype(np.float64)
    X = Q @ M[:3, :3].T + M[:3, 3]
    n = tree.query_ball_point(X, float(w.delta), return_length=True)
    hist += np.bincount(np.minimum(n, 15), minlength=16)
matched = int(hist[1:].sum())
out = {
    "workload": "bench.py C2: 50 000-point scene, 5 000-point model, delta 5 mm; 512 of its 32 768 hypotheses",
    "queries": int(hist.sum()),
    "queries_with_a_scene_point_within_delta": matched,
    "of_those_with_exactly_one": int(hist[1]),
    "of_those_with_two_or_more": int(hist[2:].sum()),
    "fraction_two_or_more": float(hist[2:].sum() / max(matched, 1)),
    "in_range_candidates_on_two_or_more_queries": float((np.arange(16) * hist)[2:].sum() / max((np.arange(16) * hist)[1:].sum(), 1)),
    "histogram_in_range_count_0_to_15plus": hist.tolist(),
    "reading": "the tie detector's second LDS operation runs per IN-RANGE candidate; a per-owner gate (>= 2 in-range candidates) would "
               "still take it for the share of in-range candidates given above -- the gate itself costs a run-mask pop-count per chunk",
}
os.makedirs(os.path.join(ROOT, "profiles", "r05_ab"), exist_ok=True)
with open(os.path.join(ROOT, "profiles", "r05_ab", "tie_gate_frequency.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out, indent=1))
