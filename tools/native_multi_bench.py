#!/usr/bin/env python3
"""Times the single-process multi-GPU scoring path behind the C ABI (pgp_multi_*, csrc/multi_gpu.hip)
on the C2 workload: n devices, 4096 hypotheses PER DEVICE per call (weak scaling, as bench.py), 8
distinct batches in rotation.  Two figures: `resident` (transforms already on the devices:
pgp_multi_score_uploaded = kernels + RCCL all-reduce + arg-max + ONE copy back + host sync per call)
and `host_pointers` (pgp_multi_score_lcp: + the pinned staging copy and H2D of the transforms).
Prints one JSON line.  bench.py runs it as a child process after its own timed region."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--devices", type=int, default=0, help="0 = every visible device")
    ap.add_argument("--mode", choices=["plain", "weighted"], default="weighted")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--hyp", type=int, default=4096, help="hypotheses per device per call")
    args = ap.parse_args()
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)   # RCCL's banner goes to C stdout
    from physimglobalpose_amd import MultiGpuScorer, LcpScorer, PGP_MODE_PLAIN, PGP_MODE_WEIGHTED, synth
    mode = PGP_MODE_PLAIN if args.mode == "plain" else PGP_MODE_WEIGHTED
    grp = MultiGpuScorer(None if args.devices <= 0 else list(range(args.devices)))
    n = grp.n_devices
    n_b = 8
    w = synth.make_workload(50000, 5000, args.hyp * n * n_b, config_id=2)
    grp.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    batches = w.T.reshape(n_b, args.hyp * n, 16)
    for b in range(3):
        grp.score(batches[b], mode, w.gate_deg)
    t0 = time.perf_counter()
    for k in range(args.steps):
        s, c, bi, bs = grp.score(batches[k % n_b], mode, w.gate_deg)
    dt_host = (time.perf_counter() - t0) / args.steps
    tim = grp.last_timing()
    grp.upload(batches[0])
    grp.score_uploaded(mode, w.gate_deg)
    t0 = time.perf_counter()
    for k in range(args.steps):
        s0, c0, bi0, bs0 = grp.score_uploaded(mode, w.gate_deg)
    dt_res = (time.perf_counter() - t0) / args.steps
    # the group's answer for batch 0 equals one device's answer for the whole batch
    one = LcpScorer(0)
    one.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    s1, c1, bi1, bs1 = one.score(batches[0], mode, w.gate_deg)
    same = bool(np.array_equal(s0, s1) and np.array_equal(c0, c1) and bi0 == bi1 and bs0 == bs1)
    out = {"devices": n, "mode": args.mode, "hypotheses_per_call": args.hyp * n, "steps": args.steps,
           "resident": {"ms_per_call": dt_res * 1e3, "hypotheses_per_s": args.hyp * n / dt_res},
           "host_pointers": {"ms_per_call": dt_host * 1e3, "hypotheses_per_s": args.hyp * n / dt_host,
                             "last_call_ms": tim},
           "equals_single_device": same,
           "path": "pgp_multi_*: one process, one host thread + stream per device, RCCL all-reduce(SUM) of "
                   "scores and counts issued from C++, arg-max on device 0"}
    os.write(real_stdout, (json.dumps(out) + "\n").encode())
    grp.close()


if __name__ == "__main__":
    main()
