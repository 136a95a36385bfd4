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


def _group(args, n_objects=1):
    from physimglobalpose_amd import MultiGpuScorer
    grp = MultiGpuScorer(None if args.devices <= 0 else list(range(args.devices)))
    for _ in range(n_objects - 1):
        grp.add_object()
    return grp


CALLS = {}   # per-call times of every _time() of this run: [min, median, max] ms, printed with the rows


def _time(fn, reps, tag=None):
    """MEDIAN seconds per call over `reps` calls timed one by one (the mean of 5-10 calls moves by a factor of three when
    one of them meets a 15 ms stall of the host; the spread goes into CALLS)."""
    import gc
    fn()
    gc.disable()   # one full pass of the cycle collector is tens of milliseconds (tools/icp_hiccup_probe.py)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = fn()
        ts.append(time.perf_counter() - t0)
    gc.enable()
    ts.sort()
    if tag:
        CALLS[tag] = [round(ts[0] * 1e3, 4), round(ts[len(ts) // 2] * 1e3, 4), round(ts[-1] * 1e3, 4)]
    return ts[len(ts) // 2], r


def objects_row(args, mode):
    """BASELINE configs[3] through the C ABI: 6 objects (20 000-point segments, 3 000-point models), 65 536 hypotheses as ONE
    flat (object, hypothesis) space over the group (pgp_multi_score_objects), against six single contexts."""
    from physimglobalpose_amd import LcpScorer, synth
    counts = [16384, 12288, 12288, 8192, 8192, 8192]
    objs = [synth.make_workload(20000, 3000, c, config_id=300 + k) for k, c in enumerate(counts)]
    grp = _group(args, 6)
    for o, w in enumerate(objs):
        grp.init_object(o, w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    Ts = [w.T for w in objs]
    reps = max(5, min(args.steps, 20))
    dt_host, got = _time(lambda: grp.score_objects(Ts, mode, 30.0), reps, "objects_host_pointers")
    grp.upload_objects(Ts)
    dt_res, got = _time(lambda: grp.score_objects_uploaded(mode, 30.0), reps, "objects_resident")
    same = True
    for w, g in zip(objs, got):
        one = LcpScorer(0)
        one.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
        a = one.score(w.T, mode, 30.0)
        same = same and bool(np.array_equal(a[0], g[0]) and np.array_equal(a[1], g[1]) and a[2:] == g[2:])
        one.close()
    grp.close()
    N = sum(counts)
    return {"objects": 6, "hypotheses": N, "ms_per_call": round(dt_res * 1e3, 4), "hypotheses_per_s": round(N / dt_res),
            "host_pointers_ms": round(dt_host * 1e3, 4), "equals_single_context": same}


def icp_row(args, mode):
    """The poses of six (segment, model) jobs -- 64 each, 2500 x 5000 points, 10 iterations from 1 mm / 0.3 degrees off --
    sharded over the group (pgp_multi_icp_refine), against one pgp_icp_refine per job on one context."""
    from physimglobalpose_amd import LcpScorer, synth
    rng = np.random.default_rng(21)
    jobs = []
    for k in range(6):
        M, _ = synth.make_model(rng, 5000)
        M = M.astype(np.float32)
        R = synth._random_rot(rng)
        t = rng.uniform(-0.2, 0.2, 3) + np.array([0, 0, 0.8])
        S = (M[rng.choice(5000, 2500, replace=False)] @ R.T + t + 0.0005 * rng.standard_normal((2500, 3))).astype(np.float32)
        Tinv = np.linalg.inv(synth._se3(R, t))
        G = np.stack([synth.colmajor16(Tinv @ synth._se3(synth._random_rot(rng, np.deg2rad(0.3)), 0.001 * rng.standard_normal(3)))
                      for _ in range(64)])
        jobs.append((S, M, G))
    grp = _group(args)
    dt, got = _time(lambda: grp.icp_refine(jobs, trim=0.9, max_iterations=10), 9, "icp_group")
    same, n_it = True, 0
    one = [LcpScorer(0) for _ in jobs]
    for sc, (S, M, G), g in zip(one, jobs, got):
        a = sc.icp_refine(S, M, G, trim=0.9, max_iterations=10)
        same = same and all(bool(np.array_equal(x, y)) for x, y in zip(a, g))
        n_it += int(a[2].sum())
    dt_one, _ = _time(lambda: [sc.icp_refine(S, M, G, trim=0.9, max_iterations=10) for sc, (S, M, G) in zip(one, jobs)], 9, "icp_one_context_per_job")
    grp.close()
    return {"jobs": 6, "poses": 6 * 64, "pose_iterations": n_it, "ms_per_call": round(dt * 1e3, 4),
            "pose_iterations_per_s": round(n_it / dt), "one_context_per_job_ms": round(dt_one * 1e3, 4), "equals_single_context": same}


def congruent_row(args, mode):
    """The bases of one object (the drop-in's case: 2043-point segment, 800-point search model, 18 682 pair-feature keys)
    sharded over the group (pgp_multi_find_congruent_batch + the fits of up to 100 quads per base), against one context."""
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _dropin import make_dropin_case
    from physimglobalpose_amd import LcpScorer
    with tempfile.TemporaryDirectory() as d:
        _, c = make_dropin_case(d)
    w, table = c["w"], c["table"]
    keys = np.array(list(table.keys()), np.int32)
    counts = np.array([len(table[tuple(k)]) for k in keys.tolist()], np.int32)
    pairs = np.concatenate([np.array(table[tuple(k)], np.int32).reshape(-1, 2) for k in keys.tolist()])
    sc = LcpScorer(0)
    sc.set_scene(w.P_xyz, w.P_nrm, w.P_w, w.delta)
    sc.set_search_model(w.Qs_xyz)
    sc.set_ppf_map(keys, counts, pairs)
    rng = np.random.default_rng(3)
    ids, inv, status = sc.select_bases(rng.random((256, 4)))
    ids, inv = ids[status == 1][:100], inv[status == 1][:100]
    base_xyz = w.P_xyz[ids]
    grp = _group(args)
    grp.init_object(0, w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    grp.set_object_search_model(0, w.Qs_xyz)
    grp.set_object_ppf_map(0, keys, counts, pairs)

    def picks_of(nq):
        return np.array([(b, j) for b in range(len(nq)) for j in range(min(int(nq[b]), 100))], np.int32).reshape(-1, 2)

    def group_call():
        nq = grp.find_congruent_batch(0, ids, base_xyz, inv, w.delta)
        return nq, grp.congruent_batch_fit(0, picks_of(nq), ids, w.centroid_P, w.centroid_Q)

    def single_call():
        nq = sc.find_congruent_batch(ids, base_xyz, inv, w.delta)
        return nq, sc.congruent_batch_fit(picks_of(nq), ids, w.centroid_P, w.centroid_Q)

    dt, (nq, fit) = _time(group_call, 15, "congruent_group")
    dt_one, (nq1, fit1) = _time(single_call, 15, "congruent_one_context")
    good = fit1[2] == 1
    same = bool(np.array_equal(nq, nq1) and np.array_equal(fit[2], fit1[2]) and np.array_equal(fit[0][good], fit1[0][good])
                and np.array_equal(fit[1][good], fit1[1][good]))
    grp.close()
    return {"bases": int(len(ids)), "quads": int(nq.sum()), "fits": int(len(fit[2])), "ms_per_call": round(dt * 1e3, 4),
            "one_context_ms": round(dt_one * 1e3, 4), "equals_single_context": same}


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
    # pre-heat: a call is ~0.15 ms and this process starts on an idle chip -- 150 ms of calls first, and never fewer
    # than 100 timed ones, whatever --steps says (bench.py passes the driver's 20)
    args.steps = max(args.steps, 100)
    t_heat = time.perf_counter()
    while time.perf_counter() - t_heat < 0.15:
        for b in range(n_b):
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
    grp.close()
    del one
    extra = {}
    for name, fn in (("objects", objects_row), ("icp_shards", icp_row), ("congruent_shards", congruent_row)):
        try:
            extra[name] = fn(args, mode)
        except Exception as e:   # a secondary row must not take the others down
            extra[name] = {"error": repr(e)}
    diag = {}
    try:   # who else is on the CPUs while this runs (bench.py starts it as a child: its own threads should be asleep)
        ppid = os.getppid()
        states = {}
        for t in os.listdir(f"/proc/{ppid}/task"):
            st = open(f"/proc/{ppid}/task/{t}/stat").read().rsplit(")", 1)[1].split()[0]
            states[st] = states.get(st, 0) + 1
        diag = {"cpus": len(os.sched_getaffinity(0)), "loadavg": os.getloadavg()[0], "parent_threads_by_state": states}
    except OSError:
        pass
    out = {"devices": n, "mode": args.mode, "hypotheses_per_call": args.hyp * n, "steps": args.steps, "diag": diag, "calls_min_median_max_ms": CALLS, **extra,
           "resident": {"ms_per_call": dt_res * 1e3, "hypotheses_per_s": args.hyp * n / dt_res},
           "host_pointers": {"ms_per_call": dt_host * 1e3, "hypotheses_per_s": args.hyp * n / dt_host,
                             "last_call_ms": tim},
           "equals_single_device": same,
           "path": "pgp_multi_*: one process, one host thread + stream per device, RCCL all-reduce(SUM) of "
                   "scores and counts issued from C++, arg-max on device 0"}
    os.write(real_stdout, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
