#!/usr/bin/env python3
"""The drop-in's own phase timings (PGP_SHIM_VERBOSE) for the in-memory path: set-up, base selection,
congruent sets, rigid fits, and what is left for scoring + bookkeeping."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _dropin import make_dropin_case  # noqa: E402

with tempfile.TemporaryDirectory() as d:
    args, case = make_dropin_case(d)
    env = dict(os.environ, PGP_SHIM_SEED="12345", SHIM_TEST_REPEAT="3", SHIM_TEST_INMEMORY="1", PGP_SHIM_VERBOSE="1")
    r = subprocess.run([os.path.join(ROOT, "shim", "test_shim")] + args, env=env, capture_output=True, text=True, timeout=600)
    print(r.stdout[-600:])
    print(r.stderr[-1500:])
