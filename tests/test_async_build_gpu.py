"""The index of a small scene is built on the context's side stream while pgp_set_scene has already returned
(csrc/grid_index.hip build_index_async): whatever reads it must wait for it by itself, and it must be the same index
the synchronous build (PGP_ASYNC_BUILD=0, every larger scene) produces.  Round 5: the build is prepared by pgp_set_scene and
queued when the caller next waits for the device or needs the index (PGP_DEFER_BUILD=0: queued at once) -- same index, and a
scene replaced before anybody asked for its index never has its build queued."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROBE = r"""
import json, sys, numpy as np, torch
sys.path.insert(0, %r)
from physimglobalpose_amd import LcpScorer, PGP_MODE_WEIGHTED, PGP_MODE_PLAIN, synth
w = synth.make_workload(2500, 700, 384, config_id=77)
sc = LcpScorer(0)
sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
sc.reserve(384)
dT = torch.from_numpy(w.T).cuda()
out = {}
for rep in range(3):
    # a new scene every time (the same points): the build of call k is queued when the scoring call below is issued
    sc.set_scene(w.P_xyz, w.P_nrm, w.P_w, w.delta)
    ds = torch.zeros(384, device="cuda"); dc = torch.zeros(384, dtype=torch.int32, device="cuda"); db = torch.zeros(2, dtype=torch.int32, device="cuda")
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        sc.score_device(dT, ds, dc, db, mode=PGP_MODE_WEIGHTED, gate_deg=w.gate_deg, stream=st)
    st.synchronize()
    out["w%%d" %% rep] = [ds.cpu().numpy().view(np.uint32).tolist(), dc.cpu().numpy().tolist(), db.cpu().numpy().tolist()]
sc.set_scene(w.P_xyz, w.P_nrm, w.P_w, w.delta)
info = sc.index_info()          # waits for the build on the host
out["info"] = {k: int(info[k]) for k in ("n_cells", "n_candidates", "n_occupied")}
s, c, bi, bs = sc.score(w.T, PGP_MODE_PLAIN)
out["plain"] = [np.asarray(s).view(np.uint32).tolist(), np.asarray(c).tolist(), int(bi)]
sc.set_scene(w.P_xyz, w.P_nrm, w.P_w, w.delta)
reg = sc.registered(w.T[int(bi)], PGP_MODE_WEIGHTED, w.gate_deg)     # straight after set_scene
out["registered"] = np.asarray(reg).tolist()
# a scene that is replaced before anybody asked for its index (its build is never queued), then the new one scored;
# and a context that is closed with a build still put off
w2 = synth.make_workload(1800, 700, 384, config_id=78)
sc.set_scene(w.P_xyz, w.P_nrm, w.P_w, w.delta)
sc.set_scene(w2.P_xyz, w2.P_nrm, w2.P_w, w2.delta)
s, c, bi, bs = sc.score(w.T, PGP_MODE_WEIGHTED, w.gate_deg)
out["replaced"] = [np.asarray(s).view(np.uint32).tolist(), np.asarray(c).tolist(), int(bi)]
fresh = LcpScorer(0)
fresh.init(w2.P_xyz, w2.P_nrm, w2.P_w, w.Q_xyz, w.Q_nrm, w2.delta)
s, c, bi, bs = fresh.score(w.T, PGP_MODE_WEIGHTED, w.gate_deg)
out["replaced_fresh"] = [np.asarray(s).view(np.uint32).tolist(), np.asarray(c).tolist(), int(bi)]
fresh.set_scene(w.P_xyz, w.P_nrm, w.P_w, w.delta)
fresh.close()
print("RESULT " + json.dumps(out))
""" % ROOT


def _run(env_extra):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, "-c", PROBE], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[7:])


def test_side_stream_build_equals_the_synchronous_build():
    a = _run({})
    b = _run({"PGP_ASYNC_BUILD": "0"})
    assert a["info"] == b["info"] and a["info"]["n_candidates"] > 0
    for k in ("w0", "w1", "w2", "plain", "registered"):
        assert a[k] == b[k], k
    assert a["w0"] == a["w1"] == a["w2"]
    assert a["replaced"] == a["replaced_fresh"] == b["replaced"]
    # the build queued by pgp_set_scene itself (PGP_DEFER_BUILD=0) instead of when the caller next waits
    c = _run({"PGP_DEFER_BUILD": "0"})
    for k in ("info", "w0", "plain", "registered", "replaced"):
        assert a[k] == c[k], k
    assert any(x != 0 for x in a["w0"][1])          # the batch registers something: the comparison is not 0 == 0
