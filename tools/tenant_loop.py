#!/usr/bin/env python3
"""One TENANT of a shared GPU: loops the two ICP forms whose workgroups wait for each other -- the clustered launch (several
workgroups per pose: the per-expansion refinement, UCTState.cpp:121-204) and the scene-sized capped form in one launch of
resident workgroups (the table alignment, SceneCfg.cpp:101,135-141) -- for `seconds`, times every call, and checks every
result against the first call's bits.  Two of these side by side on ONE device are the node beside another libpgp user (or the
segmentation CNN): tests/test_two_tenants_gpu.py.  Prints one JSON line.
usage: python tools/tenant_loop.py seconds [tag]"""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from physimglobalpose_amd import LcpScorer  # noqa: E402
from test_icp_index_gpu import _problem  # noqa: E402


def digest(res):
    h = hashlib.sha256()
    for a in res:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16]


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
    tag = sys.argv[2] if len(sys.argv) > 2 else "tenant"
    S, M, N, G = _problem(85, 5000, 2500, 48, rot_deg=4.0, trans=0.004, outliers=0.03)
    rng = np.random.default_rng(14)
    tgt = np.c_[rng.uniform(-0.6, 0.6, 60000), rng.uniform(-0.4, 0.4, 60000), 0.0005 * rng.standard_normal(60000)].astype(np.float32)
    src = (tgt[rng.choice(len(tgt), 20000, replace=False)] + np.array([0.004, -0.003, 0.002]) + 0.0008 * rng.standard_normal((20000, 3))).astype(np.float32)
    eye = np.eye(4, dtype=np.float32).T.reshape(1, 16).copy()
    kw = dict(max_iterations=30, max_corr_dist=0.01, energy_ratio=0.0, transformation_epsilon=1e-9, absolute_mse=1e-12)
    sc = LcpScorer(0)
    calls = {"clustered": lambda: sc.icp_refine(S, M, G, trim=0.9, max_iterations=12),
             "scene_sized": lambda: sc.icp_refine_ex(src, tgt, eye, **kw)}
    first = {k: digest(f()) for k, f in calls.items()}
    for f in calls.values():   # warm
        f()
    times = {k: [] for k in calls}
    bad = {k: 0 for k in calls}
    # the other tenant starts at about the same time: a barrier through the file system
    sync = os.environ.get("TENANT_SYNC")
    if sync:
        open(sync + "." + tag, "w").close()
        t_wait = time.time()
        while time.time() - t_wait < 60 and not all(os.path.exists(sync + "." + t) for t in os.environ.get("TENANT_TAGS", tag).split(",")):
            time.sleep(0.001)
    t_end = time.perf_counter() + seconds
    while time.perf_counter() < t_end:
        for k, f in calls.items():
            t0 = time.perf_counter()
            r = f()
            times[k].append(time.perf_counter() - t0)
            bad[k] += digest(r) != first[k]
    out = {"tag": tag, "seconds": seconds, "wait_ms_floor": os.environ.get("PGP_ICP_WAIT_MS", "3 (default)")}
    for k in calls:
        t = np.sort(times[k]) * 1e3
        out[k] = {"calls": len(t), "median_ms": round(float(t[len(t) // 2]), 4), "p99_ms": round(float(t[int(0.99 * len(t))]), 4),
                  "max_ms": round(float(t[-1]), 4), "over_20ms": int((t > 20).sum()), "mismatches": bad[k], "digest": first[k]}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
