"""Two PROCESSES using the library on one device: the clustered ICP launch and the scene-sized one-launch
form are plain launches of workgroups that wait for each other, co-resident only by assumption; side by side, each process can
hold a part of the device while its partners wait for the rest.  The waits' clock bounds follow the work (64 x the slowest
observed phase, 3 ms at least -- 2 s up to round 5) and a pose whose meeting ran out is ABANDONED for all its workgroups at
once, so a lost meeting costs milliseconds: every call of both tenants returns the solo run's bits, and no call takes 20 ms.
The consumers: SceneCfg.cpp:101,135-141 (table alignment) and mcts/UCTState.cpp:121-204 (refinement per expansion), in a node
that shares its GPU with the segmentation network."""
import json
import os
import subprocess
import sys
import tempfile

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(seconds, tag, env):
    return subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "tenant_loop.py"), str(seconds), tag], cwd=ROOT, env=env,
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def _result(p):
    out, err = p.communicate(timeout=300)
    assert p.returncode == 0, err[-2000:]
    return json.loads([l for l in out.splitlines() if l.startswith("{")][-1])


def test_two_tenants_on_one_device_stay_exact_and_prompt():
    env = {k: v for k, v in os.environ.items() if not k.startswith("PGP_ICP_")}
    solo = _result(_run(1.0, "solo", env))
    with tempfile.TemporaryDirectory() as d:
        env2 = dict(env, TENANT_SYNC=os.path.join(d, "go"), TENANT_TAGS="a,b")
        pa, pb = _run(5.0, "a", env2), _run(5.0, "b", env2)
        ra, rb = _result(pa), _result(pb)
    log = {"solo": solo, "tenants": [ra, rb]}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "two_tenants.json"), "w") as f:
        json.dump(log, f, indent=1)
    for r in (ra, rb):
        for form in ("clustered", "scene_sized"):
            q = r[form]
            assert q["calls"] > 50, (form, q)
            assert q["mismatches"] == 0 and q["digest"] == solo[form]["digest"], (form, q)      # the solo run's bits, every call
            assert q["p99_ms"] < 20.0 and q["over_20ms"] == 0, (form, q)                         # and promptly
