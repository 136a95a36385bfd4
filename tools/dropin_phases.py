"""Where a drop-in call's time goes (in-memory overload, the reference's real sizes: ~2000-point segment,
1500 / 800-point models, 18 682 PPF keys): PGP_SHIM_VERBOSE makes libsuper4pcs.so print one PHASES line per call;
this tool runs N calls in one process and prints median / p90 / p99 / max per phase and of the whole call.
usage: python tools/dropin_phases.py [calls=200]"""
import os, re, sys, subprocess, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _dropin import make_dropin_case
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
exe = os.path.join(ROOT, "shim", "test_shim")
with tempfile.TemporaryDirectory() as d:
    args, case = make_dropin_case(d)
    env = dict(os.environ, PGP_SHIM_SEED="12345", SHIM_TEST_REPEAT=str(n), SHIM_TEST_INMEMORY="1", PGP_SHIM_VERBOSE="1")
    r = subprocess.run([exe] + args, env=env, capture_output=True, text=True, timeout=900)
rows = []
for l in r.stderr.splitlines():
    if "PHASES" in l:
        rows.append({k: float(v) for k, v in re.findall(r"(\S+)=([\d.eE+-]+)", l)})
el = [float(x) for l in r.stdout.splitlines() if l.startswith("ELAPSED_MS") for x in l.split()[1:]]
print(f"{case['info']}; {len(rows)} calls, the first {el[0]:.1f} ms (context, code objects, PPF table upload); statistics over calls 2..{len(rows)}")
names = [k for k in rows[0] if k != "n_h"]
print(f"{'phase':20s} {'median':>8s} {'p90':>8s} {'p99':>8s} {'max':>8s}  ms")
for k in names:
    v = np.array([x[k] for x in rows[1:]])
    print(f"{k:20s} {np.median(v):8.3f} {np.percentile(v, 90):8.3f} {np.percentile(v, 99):8.3f} {v.max():8.3f}")
v = np.array(el[1:])
print(f"{'whole call':20s} {np.median(v):8.3f} {np.percentile(v, 90):8.3f} {np.percentile(v, 99):8.3f} {v.max():8.3f}   (hypotheses per call: {rows[1]['n_h']:.0f})")
worst = int(np.argmax(v)) + 1
print("slowest call", worst + 1, {k: round(rows[worst][k], 3) for k in names})
