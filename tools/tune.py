#!/usr/bin/env python3
"""A/B the scoring-kernel knobs (PGP_UNROLL, PGP_HPB) in ONE process, interleaved rounds
(cdna_hip_programming.md rule 24).  Prints median / min kernel time per variant (HIP events)."""
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from physimglobalpose_amd import LcpScorer, synth, PGP_MODE_PLAIN, PGP_MODE_WEIGHTED  # noqa: E402


def main():
    unrolls = [int(x) for x in os.environ.get("TUNE_UNROLL", "0,2").split(",")]
    hpbs = [int(x) for x in os.environ.get("TUNE_HPB", "0").split(",")]
    modes = os.environ.get("TUNE_MODES", "plain,weighted").split(",")
    rounds = int(os.environ.get("TUNE_ROUNDS", "7"))
    n_h = int(os.environ.get("TUNE_NH", "4096"))
    w = synth.make_workload(50000, 5000, n_h, config_id=2)
    dT = torch.from_numpy(w.T).cuda()
    ds = torch.zeros(n_h, device="cuda")
    dc = torch.zeros(n_h, dtype=torch.int32, device="cuda")
    db = torch.zeros(2, dtype=torch.int32, device="cuda")
    variants = {}
    for u, hpb in itertools.product(unrolls, hpbs):
        os.environ["PGP_UNROLL"] = str(u)
        os.environ["PGP_HPB"] = str(hpb)
        sc = LcpScorer(0)
        sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
        sc.reserve(n_h)
        sc.set_kernel_timing(True)
        variants[(u, hpb)] = sc
    ref = {}
    res = {(k, m): [] for k in variants for m in modes}
    for r in range(rounds + 1):
        for k, sc in variants.items():
            for m in modes:
                mode = PGP_MODE_PLAIN if m == "plain" else PGP_MODE_WEIGHTED
                sc.kernel_timing(reset=True)
                for _ in range(10):
                    sc.score_device(dT, ds, dc, db, mode=mode)
                torch.cuda.synchronize()
                n, ms = sc.kernel_timing(reset=True)
                if r:  # round 0 = warm-up
                    res[(k, m)].append(ms / n * 1e3)
                c = dc.cpu().numpy().copy()
                if m not in ref:
                    ref[m] = c
                assert np.array_equal(ref[m], c), (k, m)
    for (k, m), v in sorted(res.items(), key=lambda kv: (kv[0][1], np.median(kv[1]))):
        print(f"{m:9s} unroll={k[0]} hpb={k[1]:3d}  median {np.median(v):8.1f} us  min {np.min(v):8.1f} us"
              f"  -> {n_h / np.median(v):.2f} M hyp/s")


if __name__ == "__main__":
    main()
