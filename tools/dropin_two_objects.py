"""Two objects alternating through the drop-in in one process (the node's object loop): per-call time with the
per-object context cache against PGP_SHIM_NO_CACHE=1 (everything re-sent on every call).  usage: python tools/dropin_two_objects.py"""
import os, sys, subprocess, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _dropin import make_dropin_case
exe = os.path.join(ROOT, "shim", "test_shim")
with tempfile.TemporaryDirectory() as d:
    args, case = make_dropin_case(d)
    for label, extra in (("one object, cached", {}), ("two objects alternating, per-object contexts", {"SHIM_TEST_TWO_OBJECTS": "1"}),
                         ("two objects alternating, PGP_SHIM_NO_CACHE=1", {"SHIM_TEST_TWO_OBJECTS": "1", "PGP_SHIM_NO_CACHE": "1"})):
        env = dict(os.environ, PGP_SHIM_SEED="12345", SHIM_TEST_REPEAT="100", SHIM_TEST_INMEMORY="1", **extra)
        r = subprocess.run([exe] + args, env=env, capture_output=True, text=True, timeout=900)
        ms = np.array([float(x) for l in r.stdout.splitlines() if l.startswith("ELAPSED_MS") for x in l.split()[1:]])
        best = [l for l in r.stdout.splitlines() if l.startswith("BEST_SCORE")]
        print(f"{label:50s}: median {np.median(ms[4:]):.3f} ms, p99 {np.percentile(ms[4:], 99):.3f}, calls 1-4 {ms[:4].round(1).tolist()}  {best[0] if best else r.stderr[-200:]}")
