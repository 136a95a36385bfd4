#!/usr/bin/env python3
"""The candidate-format experiment (csrc/lcp_score.hip PGP_CAND8: fp16 offsets from the cell centre + 16-bit ids, 8 bytes
per candidate) against the exact kernel on the C2 batch: how many inlier counts / weighted scores move (the offsets carry
~4 um of error: decisions next to the radius can flip), and the step time of both at several cell edges.
usage (GPU box): python tools/cand8_check.py   -- runs itself under PGP_LIB for both libraries"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "--worker":
    import time
    import numpy as np
    import torch
    from physimglobalpose_amd import LcpScorer, synth, PGP_MODE_PLAIN, PGP_MODE_WEIGHTED
    w = synth.make_workload(50000, 5000, 4096, config_id=2)
    sc = LcpScorer(0)
    sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    sc.reserve(4096)
    info = sc.index_info()
    dT = torch.from_numpy(w.T).cuda()
    ds = torch.zeros(4096, device="cuda"); dc = torch.zeros(4096, dtype=torch.int32, device="cuda"); db = torch.zeros(2, dtype=torch.int32, device="cuda")
    out = {"cell_size": float(info["cell_size"]), "n_candidates": int(info["n_candidates"]), "bytes_index": int(info["bytes_index"])}
    for mode, name in ((PGP_MODE_PLAIN, "plain"), (PGP_MODE_WEIGHTED, "weighted")):
        best = []
        for rep in range(5):
            for _ in range(20):
                sc.score_device(dT, ds, dc, db, mode=mode)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                sc.score_device(dT, ds, dc, db, mode=mode)
            torch.cuda.synchronize()
            best.append((time.perf_counter() - t0) / 200 * 1e6)
        out[name + "_step_us"] = min(best)
        out[name + "_counts"] = dc.cpu().numpy().tolist()
        out[name + "_scores"] = ds.cpu().numpy().tolist()
        out[name + "_best"] = int(db[0])
    print("RESULT " + json.dumps(out))
    sys.exit(0)

import numpy as np
rows = {}
for ratio in ("0.85", "0.7", "0.6", "0.5"):
    for name, lib in (("float4", os.path.join(ROOT, "physimglobalpose_amd", "libpgp.so")), ("cand8", os.path.join(ROOT, "tools", "ab", "libpgp_cand8.so"))):
        env = dict(os.environ, PGP_LIB=lib, PGP_CELL_RATIO=ratio)
        r = subprocess.run([sys.executable, __file__, "--worker"], env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print(name, ratio, "FAILED", r.stderr[-500:])
            continue
        rows[(name, ratio)] = json.loads(line[-1][7:])
out = {}
for ratio in ("0.85", "0.7", "0.6", "0.5"):
    a, b = rows.get(("float4", ratio)), rows.get(("cand8", ratio))
    if not a or not b:
        continue
    ca, cb = np.array(a["plain_counts"]), np.array(b["plain_counts"])
    sa, sb = np.array(a["weighted_scores"]), np.array(b["weighted_scores"])
    out[ratio] = {"cell_size_mm": a["cell_size"] * 1e3, "n_candidates": a["n_candidates"], "bytes_index_float4": a["bytes_index"],
                  "plain_step_us": [a["plain_step_us"], b["plain_step_us"]], "weighted_step_us": [a["weighted_step_us"], b["weighted_step_us"]],
                  "hypotheses_whose_inlier_count_moved": int((ca != cb).sum()), "largest_count_change": int(np.abs(ca - cb).max()),
                  "largest_weighted_score_change": float(np.abs(sa - sb).max()), "same_best": a["weighted_best"] == b["weighted_best"]}
    print(ratio, json.dumps(out[ratio]))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"how": "tools/cand8_check.py: [float4, cand8] step times (min of 5 x 200 steps, one C2 batch) per PGP_CELL_RATIO", "ratios": out},
          open(os.path.join(ROOT, "gpurun_out", "cand8_check.json"), "w"), indent=1)
