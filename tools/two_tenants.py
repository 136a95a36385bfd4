#!/usr/bin/env python3
"""profiles/r06_ab/two_tenants.log: tools/tenant_loop.py twice side by side on one device (and once beside a process that keeps
the device busy with matrix products), with this round's clock bounds (3 ms floor, scaled by the observed phases, abandon word)
and with round 5's 2 s (PGP_ICP_WAIT_MS=2000).  usage: python tools/two_tenants.py [seconds]"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sec = sys.argv[1] if len(sys.argv) > 1 else "5"
HOG = ("import torch, time, sys\n"
       "a = torch.randn(8192, 8192, device='cuda'); b = a\n"
       "t = time.time()\n"
       "while time.time() - t < float(sys.argv[1]) + 3:\n"
       "    for _ in range(20): b = torch.sin(b @ a * 1e-4)\n"
       "    torch.cuda.synchronize()\n")


def run(tags, env_extra, hog=False):
    env = {k: v for k, v in os.environ.items() if not k.startswith("PGP_ICP_")}
    env.update(env_extra)
    with tempfile.TemporaryDirectory() as d:
        env2 = dict(env, TENANT_SYNC=os.path.join(d, "go"), TENANT_TAGS=",".join(tags))
        ps = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "tenant_loop.py"), sec, t], env=env2, stdout=subprocess.PIPE,
                               stderr=subprocess.PIPE, text=True) for t in tags]
        h = subprocess.Popen([sys.executable, "-c", HOG, sec], env=env) if hog else None
        outs = [p.communicate(timeout=600) for p in ps]
        if h:
            h.wait(timeout=600)
    res = []
    for (o, e), p in zip(outs, ps):
        lines = [l for l in o.splitlines() if l.startswith("{")]
        res.append(json.loads(lines[-1]) if lines else {"error": e[-400:]})
    return res


for name, tags, env, hog in (("two tenants, this round's bounds", ["a", "b"], {}, False),
                             ("two tenants, round 5's 2 s bound", ["a", "b"], {"PGP_ICP_WAIT_MS": "2000"}, False),
                             ("one tenant beside a matrix-product hog, this round's bounds", ["a"], {}, True),
                             ("one tenant beside a matrix-product hog, round 5's 2 s bound", ["a"], {"PGP_ICP_WAIT_MS": "2000"}, True)):
    print("#", name, flush=True)
    for r in run(tags, env, hog):
        print(json.dumps(r), flush=True)
