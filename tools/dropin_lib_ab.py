#!/usr/bin/env python3
"""A/B of the drop-in on ONE box between two builds of libpgp.so: the tree's and tools/ab/before/libpgp.so (LD_LIBRARY_PATH
comes before the shim's RUNPATH), alternating, same files, same seed.  usage: python tools/dropin_lib_ab.py [rounds]"""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _dropin import make_dropin_case
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
exe = os.path.join(ROOT, "shim", "test_shim")
before = os.path.join(ROOT, "tools", "ab", "before")
with tempfile.TemporaryDirectory() as d:
    args, case = make_dropin_case(d)
    for r in range(rounds):
        for name, lib in (("tree", None), ("before", before)):
            for mode, extra, calls in (("memory", {"SHIM_TEST_INMEMORY": "1"}, 200), ("file", {}, 60)):
                env = dict(os.environ, PGP_SHIM_SEED="12345", SHIM_TEST_REPEAT=str(calls), **extra)
                if lib:
                    env["LD_LIBRARY_PATH"] = lib + ":" + env.get("LD_LIBRARY_PATH", "")
                out = subprocess.run([exe] + args, env=env, capture_output=True, text=True, timeout=600)
                ms = [float(x) for l in out.stdout.splitlines() if l.startswith("ELAPSED_MS") for x in l.split()[1:]]
                if len(ms) < 3:
                    print(name, mode, "failed:", out.stdout[-300:], out.stderr[-300:])
                    continue
                rest = np.array(ms[1:])
                print(f"round {r} {name:6s} {mode:6s}: median {np.median(rest):.3f} ms, p90 {np.percentile(rest, 90):.3f}, min {rest.min():.3f}", flush=True)
