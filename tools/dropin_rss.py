#!/usr/bin/env python3
"""Does a long run of drop-in calls grow?  Peak resident set of shim/test_shim after 300 and after 3000 calls (in-memory
overload and file hand-off, fixed seed): a leak per call would show as a difference that scales with the calls."""
import os, resource, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _dropin import make_dropin_case
exe = os.path.join(ROOT, "shim", "test_shim")
code = "import resource, subprocess, sys; subprocess.run(sys.argv[1:], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL); print(resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss)"
with tempfile.TemporaryDirectory() as d:
    args, _ = make_dropin_case(d)
    for mode in ({"SHIM_TEST_INMEMORY": "1"}, {}):
        peak = {}
        for n in (300, 3000):
            env = dict(os.environ, PGP_SHIM_SEED="12345", SHIM_TEST_REPEAT=str(n), **mode)
            r = subprocess.run([sys.executable, "-c", code, exe] + args, env=env, capture_output=True, text=True, timeout=900)
            peak[n] = int(r.stdout.strip().splitlines()[-1])
        name = "in-memory" if mode else "file hand-off"
        print(f"{name}: peak RSS {peak[300] / 1024:.1f} MB after 300 calls, {peak[3000] / 1024:.1f} MB after 3000 "
              f"({(peak[3000] - peak[300]) / 2700:.2f} KB per call)", flush=True)
