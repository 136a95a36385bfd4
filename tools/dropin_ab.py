#!/usr/bin/env python3
"""A/B of the drop-in's file hand-off on ONE box: shim/test_shim (this tree) against tools/ab/shim_old/test_shim (the shim of
another commit built beside it), alternating, same files, same seed.  usage: python tools/dropin_ab.py [rounds]"""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _dropin import make_dropin_case
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
with tempfile.TemporaryDirectory() as d:
    args, case = make_dropin_case(d)
    for r in range(rounds):
        for name, exe in (("new", os.path.join(ROOT, "shim", "test_shim")), ("old", os.path.join(ROOT, "tools", "ab", "shim_old", "test_shim"))):
            for mode, extra, calls in (("file", {}, 40), ("memory", {"SHIM_TEST_INMEMORY": "1"}, 100)):
                env = dict(os.environ, PGP_SHIM_SEED="12345", SHIM_TEST_REPEAT=str(calls), **extra)
                out = subprocess.run([exe] + args, env=env, capture_output=True, text=True, timeout=600)
                ms = [float(x) for l in out.stdout.splitlines() if l.startswith("ELAPSED_MS") for x in l.split()[1:]]
                rest = np.array(ms[1:])
                print(f"round {r} {name} {mode}: median {np.median(rest):.3f} ms, p90 {np.percentile(rest, 90):.3f}, min {rest.min():.3f}", flush=True)
