import os, sys, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _dropin import make_dropin_case
exe = os.path.join(ROOT, "shim", "test_shim")
with tempfile.TemporaryDirectory() as d:
    args, case = make_dropin_case(d)
    env = dict(os.environ, PGP_SHIM_SEED="12345", SHIM_TEST_REPEAT="4", SHIM_TEST_INMEMORY="1", PGP_SHIM_VERBOSE="1")
    r = subprocess.run([exe] + args, env=env, capture_output=True, text=True, timeout=600)
    print(r.stdout[-600:]); print(r.stderr[-2500:])
