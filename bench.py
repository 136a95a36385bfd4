#!/usr/bin/env python3
"""bench.py -- pose hypotheses/sec LCP-scored (BASELINE.json metric) on MI355X.

A "step" = one pass of the hot path over one batch: every rank LCP-scores its shard of
hypotheses (C2: 4096 per GPU, 5 000-pt model vs 50 000-pt scene, plain LCP = the reference's
Verify without early-out) with the clouds, the index and the transforms already resident in HBM,
then (N > 1) the per-hypothesis scores are combined with one RCCL all-reduce and the arg-max is
taken.  value = hypotheses all ranks scored / max-over-ranks wall time.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode plain|weighted]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the round prompt): metric/value/unit/... plus
"roofline" (dominant kernel, algorithmic bytes / HIP-event duration vs the 8 TB/s HBM peak) and
"cpu_baseline" (the CPU oracle timed on this box's host cores, rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

N_SCENE, N_MODEL, N_HYP = 50000, 5000, 4096   # BASELINE.json configs[1] (C2)
TIMING_STRIDE = 8   # every 8th launch of the timed region carries HIP events
BUCKET = 8          # steps whose score vectors share one all-reduce (N > 1)


def algorithmic_bytes_per_hypothesis(n_scene, n_model, mode):
    """plain 12|Q|+12|P|+52, weighted 24|Q|+28|P|+52."""
    if mode == "plain":
        return 12 * n_model + 12 * n_scene + 52
    return 24 * n_model + 28 * n_scene + 52


def usable_cpus(n_threads_max):
    """Threads worth starting: the CPUs this process may run on, capped by the cgroup CPU quota (the
    GPU box shows 256 logical CPUs but grants a quota of 16 -- more threads than that only add
    throttling) and by what the OpenMP runtime offers."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    note = f"{n} schedulable CPUs"
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            q = max(1, int(-(-int(quota) // int(period))))
            if q < n:
                n, note = q, f"cgroup CPU quota {q} of {n} logical CPUs"
    except (OSError, ValueError):
        pass
    return max(1, min(n, n_threads_max)), note


def cpu_baseline(w, mode, budget_s=8.0):
    """Time the CPU path on this box's cores.  The C restatement (oracle/pgp_oracle.c, OpenMP) is
    always there ("port"); where the prebuilt oracle/_ref/libpgp_ref.so travelled along, the harness
    over the reference's own kd-tree header is timed as well, one instance per host thread (the
    reference's KdTree is not re-entrant, kdtree.h:311), and reported as the headline CPU number
    ("reference") since it is the faster of the two."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _checkers import Oracle, oracle_lib
    cores, cores_note = usable_cpus(int(oracle_lib().orc_max_threads()))
    orc = Oracle(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm)
    m = 0 if mode == "plain" else 1
    # warm the thread pool up (the first parallel regions after an idle spell run far below
    # speed on shared hosts), calibrate, then size the sample for ~budget_s of wall time
    orc.score_batch(w.T[:1024], w.delta, mode=m, gate_deg=w.gate_deg, threads=cores)
    t0 = time.perf_counter()
    orc.score_batch(w.T[:1024], w.delta, mode=m, gate_deg=w.gate_deg, threads=cores)
    rate = 1024 / max(time.perf_counter() - t0, 1e-6)
    n = int(min(max(rate * budget_s, 256), 64 * len(w.T)))
    reps = -(-n // len(w.T))
    T = np.concatenate([w.T] * reps)[:n]
    t0 = time.perf_counter()
    orc.score_batch(T, w.delta, mode=m, gate_deg=w.gate_deg, threads=cores)
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    n1 = max(64, min(512, int(rate / max(cores, 1) * 3)))
    orc.score_batch(w.T[:n1], w.delta, mode=m, gate_deg=w.gate_deg, threads=1)
    dt1 = time.perf_counter() - t0
    port = {
        "value": n / dt, "unit": "hypotheses/s", "cores": cores, "kind": "port",
        "sample": f"{n} hypotheses of the same C2 batch (cycled), kd-tree oracle "
                  f"(oracle/pgp_oracle.c, OpenMP x{cores}; {cores_note}), {dt:.1f} s",
        "one_thread_value": n1 / dt1,
    }
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libpgp_ref.so")):
        return port
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor
    from _checkers import Ref, ref_lib, _fp
    L = ref_lib()
    L.ref_score_batch.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_float, C.c_int, C.POINTER(C.c_float)]

    def run(ref, Ts):   # one foreign call per thread: ctypes drops the GIL for its duration
        out = np.zeros(len(Ts), np.float32)
        L.ref_score_batch(ref.h, _fp(Ts), len(Ts), C.c_float(w.delta), m, _fp(out))
        return out

    refs = [Ref(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm) for _ in range(cores)]
    t0 = time.perf_counter()
    s_ref = run(refs[0], w.T[:256])
    r1 = 256 / (time.perf_counter() - t0)
    s_port, _, _ = orc.score_batch(w.T[:256], w.delta, mode=m, gate_deg=w.gate_deg, threads=cores)
    with ThreadPoolExecutor(cores) as ex:
        # calibrate the ALL-THREAD rate first (the box's logical cores are shared: 128 threads run
        # far below 128 x the one-thread rate), then size the sample for ~budget_s of wall time
        t0 = time.perf_counter()
        list(ex.map(lambda r: run(r, w.T[:32]), refs))
        agg = 32 * cores / (time.perf_counter() - t0)
        per = int(min(max(agg * budget_s / cores, 32), 16 * len(w.T)))
        Ts = np.ascontiguousarray(np.concatenate([w.T] * (-(-per // len(w.T))))[:per])
        t0 = time.perf_counter()
        list(ex.map(lambda r: run(r, Ts), refs))
        dtr = time.perf_counter() - t0
    return {
        "value": per * cores / dtr, "unit": "hypotheses/s", "cores": cores, "kind": "reference",
        "sample": f"{per} hypotheses of the same C2 batch per thread x {cores} threads, one instance of the "
                  f"reference kd-tree (oracle/_ref, kdtree.h + Eigen loop bodies) per thread; {cores_note}; {dtr:.1f} s",
        "one_thread_value": r1,
        "agrees_with_port": bool(np.array_equal(s_ref, s_port)),
        "port": port,
    }


def other_rows(sc, w, torch):
    """Secondary measurements for the other rows of SURVEY section 8 (not the headline metric):
    weighted LCP, batched ICP, congruent-set extraction, rigid fits.  Device time via host wall
    clock around synchronous C-ABI calls, inputs staged per call (PCIe inclusive)."""
    from physimglobalpose_amd import PGP_MODE_WEIGHTED, synth
    out = {}
    rng = np.random.default_rng(0)

    def timed(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            r = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps, r

    # the live mode of the reference (operMode 1 -> WeightedVerify): same batch, same clouds
    dT = torch.from_numpy(w.T[:N_HYP]).cuda()
    ds = torch.zeros(N_HYP, device="cuda")
    dc = torch.zeros(N_HYP, dtype=torch.int32, device="cuda")
    db = torch.zeros(2, dtype=torch.int32, device="cuda")
    for _ in range(20):
        sc.score_device(dT, ds, dc, db, mode=PGP_MODE_WEIGHTED, gate_deg=w.gate_deg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        sc.score_device(dT, ds, dc, db, mode=PGP_MODE_WEIGHTED, gate_deg=w.gate_deg)
    torch.cuda.synchronize()
    dtw = (time.perf_counter() - t0) / 200
    out["weighted_lcp"] = {"hypotheses_per_s": N_HYP / dtw, "ms_per_step": dtw * 1e3, "gate_deg": float(w.gate_deg),
                           "algorithmic_bytes_per_hypothesis": algorithmic_bytes_per_hypothesis(N_SCENE, N_MODEL, "weighted")}
    # ICP: 64 poses, 2500-pt segment vs 5000-pt model, 10 iterations each (trim 0.9)
    seg = w.Q_xyz[rng.choice(len(w.Q_xyz), 2500, replace=False)]
    R = synth._rot_axis_angle([0.2, 0.5, -0.4], 0.8)
    S = (seg @ R.T + np.array([0.1, 0.0, 0.7])).astype(np.float32)
    Tinv = np.linalg.inv(synth._se3(R, np.array([0.1, 0.0, 0.7])))
    G = np.stack([synth.colmajor16(Tinv @ synth._se3(synth._random_rot(rng, np.deg2rad(5)), 0.005 * rng.standard_normal(3)))
                  for _ in range(64)])
    dt, (_, _, its) = timed(lambda: sc.icp_refine(S, w.Q_xyz, G, trim=0.9, max_iterations=10), reps=3)
    n_it = int(its.sum())
    out["icp"] = {"poses": 64, "n_src": 2500, "n_tgt": len(w.Q_xyz), "iterations_total": n_it,
                  "pose_iterations_per_s": n_it / dt, "ms_per_call": dt * 1e3,
                  "algorithmic_GBps": n_it * (12 * 2500 + 12 * len(w.Q_xyz) + 112) / dt / 1e9}
    # congruent sets on a 1000-pt search model
    w2 = synth.make_workload(4000, 2000, 4, config_id=3, n_search=1000)
    sc.set_search_model(w2.Qs_xyz)
    T = w2.T_gt.reshape(4, 4).T
    ids = rng.choice(1000, 4, replace=False)
    base = (w2.Qs_xyz[ids] @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
    d1 = float(np.linalg.norm(base[0] - base[1]))
    d6 = float(np.linalg.norm(base[2] - base[3]))
    dt1, p1 = timed(lambda: sc.extract_pairs(d1, w.delta, cap=1 << 20))
    p6 = sc.extract_pairs(d6, w.delta, cap=1 << 20)
    dt2, q = timed(lambda: sc.find_congruent(base, 0.4, 0.6, w.delta, p1, p6, cap=1 << 20))
    out["congruent"] = {"n_search": 1000, "pairs": int(len(p1)), "pairs_per_s": len(p1) / dt1,
                        "pair_tests_per_s": 1000 * 999 / 2 / dt1, "quads": int(len(q)),
                        "pair_pairs_per_s": float(len(p1)) * len(p6) / dt2, "ms_extract": dt1 * 1e3, "ms_find": dt2 * 1e3}
    # rigid fits: 10 000 (base, quad) pairs
    sc.set_search_model(w.Qs_xyz)
    b = rng.integers(0, len(w.P_xyz), (10000, 4)).astype(np.int32)
    qd = rng.integers(0, len(w.Qs_xyz), (10000, 4)).astype(np.int32)
    dt3, _ = timed(lambda: sc.rigid_from_congruent(b, qd, w.centroid_P, w.centroid_Q))
    out["rigid_fit"] = {"pairs": 10000, "fits_per_s": 10000 / dt3, "ms_per_call": dt3 * 1e3}
    # MCTS leaf cost: 64 rendered 640 x 480 depth images against one observed image (host pointers)
    obs = rng.uniform(0.4, 1.2, (480, 640)).astype(np.float32)
    ren = (obs[None] + rng.normal(0, 0.02, (64, 480, 640))).astype(np.float32)
    dt5, _ = timed(lambda: sc.depth_cost(obs, ren, 0.01), reps=3)
    out["depth_cost"] = {"images": 64, "pixels": 640 * 480, "images_per_s": 64 / dt5, "ms_per_call": dt5 * 1e3,
                         "note": "host pointers: 79 MB of rendered depth cross PCIe per call"}
    # depth image -> segment cloud (decode + mask + back-projection, ordered compaction)
    raw = rng.integers(2000, 60000, (480, 640)).astype(np.uint16)
    msk = (rng.random((480, 640)) < 0.5).astype(np.uint8)
    Kc = np.array([[614.0, 0, 322.5], [0, 614.0, 239.7], [0, 0, 1]], np.float32)
    dt6, cloud = timed(lambda: sc.backproject_depth(raw, Kc, msk))
    out["backproject"] = {"pixels": 640 * 480, "points": int(len(cloud)), "ms_per_call": dt6 * 1e3}
    # greedy clustering of the C2 batch by its own weighted scores (all 4096 admitted: fraction 0)
    sw, _, _, bs = sc.score(w.T, PGP_MODE_WEIGHTED, w.gate_deg)
    dt4, (rep, _) = timed(lambda: sc.cluster_poses(w.T, sw + np.float32(1e-6), bs, accept_fraction=0.0))
    m = len(w.T)
    out["cluster"] = {"poses": m, "clusters": int(len(rep)), "pair_tests_per_s": m * (m - 1) / 2 / dt4,
                      "ms_per_call": dt4 * 1e3}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--mode", choices=["plain", "weighted"], default="plain")
    ap.add_argument("--hyp", type=int, default=N_HYP, help="hypotheses per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    # The contract is ONE JSON line on stdout.  Libraries below us write there too (RCCL prints a
    # version banner through C stdio, flushed at exit, i.e. AFTER anything Python printed): keep the
    # real stdout aside, point fd 1 at stderr for the whole run, and write the line to the saved
    # descriptor at the very end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    # one rank per GPU; PGP_DIST_BACKEND=gloo + fewer GPUs than ranks is a functional smoke mode
    # (ranks share a device, the collective goes through the host) used to exercise the N > 1
    # code path on a 1-GPU box -- never a performance configuration
    backend = os.environ.get("PGP_DIST_BACKEND", "nccl")
    dev_index = local_rank % max(torch.cuda.device_count(), 1) if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # PGP_BENCH_FORCE_DIST=1: take the collective path even with one rank (exercises RCCL init,
    # the asynchronous all-reduce and the stream-level wait on a 1-GPU box)
    multi = world > 1 or os.environ.get("PGP_BENCH_FORCE_DIST") == "1"
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from physimglobalpose_amd import LcpScorer, PGP_MODE_PLAIN, PGP_MODE_WEIGHTED, synth

    mode = PGP_MODE_PLAIN if args.mode == "plain" else PGP_MODE_WEIGHTED
    n_h = args.hyp
    # every rank builds the same scene/model (replicated) and takes its own slice of
    # a world*n_h hypothesis batch (weak scaling: per-GPU work fixed)
    w = synth.make_workload(N_SCENE, N_MODEL, n_h * world, config_id=2)
    sc = LcpScorer(dev_index)
    t0 = time.perf_counter()
    sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    cold_ms = (time.perf_counter() - t0) * 1e3
    sc.reserve(n_h)
    T_all = torch.from_numpy(w.T).to(dev)
    d_T = T_all[rank * n_h:(rank + 1) * n_h].contiguous()
    # Bucketed exchange: the score vectors of BUCKET consecutive steps share one all-reduce (one
    # collective launch costs the scoring stream ~11 us -- tools/dist_overhead.py -- which is 12 % of
    # a 95 us step; per-step messages are 4 B x world x n_h, far below the bandwidth regime).  Two
    # buckets alternate: the collective of one runs on RCCL's stream while the scoring kernels
    # fill the other.  Each step's vector is still all-reduced in full and arg-maxed locally.
    n_buf = 2 if multi else 1
    n_slot = BUCKET if multi else 1
    bufs = [torch.zeros(n_slot, world * n_h, dtype=torch.float32, device=dev) for _ in range(n_buf)]
    works = [None] * n_buf
    d_scores_all = bufs[0][0]
    d_scores = d_scores_all[rank * n_h:(rank + 1) * n_h]
    d_counts = torch.zeros(n_h, dtype=torch.int32, device=dev)
    d_best = torch.zeros(2, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev)
    state = {"k": 0, "argmax": None, "open": None}

    # The exchange's tail (wait for the collective, arg-max, re-zeroing the bucket) runs on its own
    # stream, beside the next steps' scoring kernels: the main stream carries scoring only and
    # waits, once per bucket, on an event recorded a bucket earlier.
    post = torch.cuda.Stream(dev) if multi else None
    ready = [torch.cuda.Event() for _ in range(n_buf)] if multi else []

    def finish(b):
        """Complete the exchange that was started on bucket b (call with `post` current): every rank
        then holds all scores of the bucket's steps (north_star: "RCCL all-reduce over xGMI of the
        per-hypothesis LCP scores") and takes each step's arg-max locally."""
        if works[b] is not None:
            works[b].wait()          # stream-level wait under nccl; host wait under gloo
            works[b] = None
            state["argmax"] = torch.argmax(bufs[b], dim=1)

    def exchange(b):
        works[b] = dist.all_reduce(bufs[b], op=dist.ReduceOp.SUM, async_op=True)
        state["open"] = None

    def step():
        if not multi:
            sc.score_device(d_T, d_scores, d_counts, d_best, mode=mode, gate_deg=w.gate_deg, stream=stream)
            return
        k = state["k"]
        state["k"] += 1
        b, j = (k // BUCKET) % n_buf, k % BUCKET
        if j == 0:                   # a new bucket: its previous contents must be consumed and cleared
            with torch.cuda.stream(post):
                finish(b)
                bufs[b].zero_()      # every rank fills only its slice of a zeroed vector: sum == gather
                ready[b].record(post)
            stream.wait_event(ready[b])
            state["open"] = b
        sc.score_device(d_T, bufs[b][j, rank * n_h:(rank + 1) * n_h], d_counts, d_best, mode=mode,
                        gate_deg=w.gate_deg, stream=stream)
        state["last"] = (b, j)
        if j == BUCKET - 1:
            exchange(b)

    def drain():
        if multi:
            if state["open"] is not None:        # a partly filled bucket: exchange what it holds
                exchange(state["open"])
            with torch.cuda.stream(post):
                for b in range(n_buf):
                    finish(b)
            stream.wait_stream(post)
            state["k"] = 0                       # the next step starts a fresh bucket

    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    # HIP events on every 8th launch of the timed region: a timed dispatch costs the stream ~8 us,
    # so timing all of them would take 7 % off the throughput being measured
    sc.set_kernel_timing(TIMING_STRIDE)
    sc.kernel_timing(reset=True)
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_issue = time.perf_counter() - t0   # host time to ENQUEUE the steps (the GPU may still be busy)
    drain()                          # every step's exchange and arg-max complete inside the timed region
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    launches, kern_ms = sc.kernel_timing(reset=True)
    sc.set_kernel_timing(False)
    if multi:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # sanity inside the bench: the device result of the last step matches a host-pointer call
    best = d_best.cpu().numpy()
    s_host, _, bi_host, _ = sc.score(w.T[rank * n_h:(rank + 1) * n_h], mode, w.gate_deg)
    last = bufs[state["last"][0]][state["last"][1]] if multi else d_scores_all
    assert np.array_equal(s_host, last[rank * n_h:(rank + 1) * n_h].cpu().numpy()) and bi_host == int(best[0])
    if multi:
        # the combined vector of the last step: every slice present, arg-max = the global best
        s_all = last.cpu().numpy()
        assert (s_all.reshape(world, n_h).max(axis=1) > 0).all()
        assert int(torch.argmax(last)) == int(np.argmax(s_all))

    if rank == 0:
        total_h = n_h * world * args.steps
        value = total_h / dt
        B_h = algorithmic_bytes_per_hypothesis(N_SCENE, N_MODEL, args.mode)
        kern_avg_ms = kern_ms / max(launches, 1)
        achieved = B_h * n_h / (kern_avg_ms * 1e-3) / 1e9 if launches else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(f"score_hypotheses_{args.mode}_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "pose hypotheses/sec LCP-scored (50k-pt scene x 5k-pt model)",
            "value": value, "unit": "hypotheses/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "host_issue_ms_per_step": t_issue / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C2 (BASELINE.json configs[1]): 1 object, 5000-pt model vs "
                                   "50000-pt synthetic scene, 4096 hypotheses per GPU per step, "
                                   f"{args.mode} LCP, delta 5 mm",
                       "n_scene": N_SCENE, "n_model": N_MODEL, "hypotheses_per_gpu": n_h,
                       "mode": args.mode, "sharding": f"hypotheses x{world}, clouds replicated",
                       "exchange": (f"one all-reduce(SUM) per {BUCKET} steps (bucketed), overlapped with scoring"
                                    if multi else "none (one rank)")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": f"score_hypotheses<{args.mode}>", "launches": launches,
                         "timed_every": TIMING_STRIDE,
                         "avg_kernel_ms": kern_avg_ms, "algorithmic_bytes_per_hypothesis": B_h},
            "index": sc.index_info(), "cold_setup_ms": cold_ms,
            "best_index": int(best[0]),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w, args.mode)
            try:
                out["other_rows"] = other_rows(sc, w, torch)
            except Exception as e:  # secondary numbers must never take the headline line down
                out["other_rows"] = {"error": repr(e)}
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if multi:
        dist.destroy_process_group()
    os.close(real_stdout)


if __name__ == "__main__":
    main()
