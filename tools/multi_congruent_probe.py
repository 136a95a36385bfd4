"""Where the time of the device group's congruent-set calls goes (one device): each call of tools/native_multi_bench.py's
congruent row timed on its own, group against single context.
usage: python tools/multi_congruent_probe.py [devices]"""
import os, sys, time, tempfile, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from _dropin import make_dropin_case
from physimglobalpose_amd import LcpScorer, MultiGpuScorer

n_dev = int(sys.argv[1]) if len(sys.argv) > 1 else 1
with tempfile.TemporaryDirectory() as d:
    _, c = make_dropin_case(d)
w, table = c["w"], c["table"]
keys = np.array(list(table.keys()), np.int32)
counts = np.array([len(table[tuple(k)]) for k in keys.tolist()], np.int32)
pairs = np.concatenate([np.array(table[tuple(k)], np.int32).reshape(-1, 2) for k in keys.tolist()])
sc = LcpScorer(0)
sc.set_scene(w.P_xyz, w.P_nrm, w.P_w, w.delta)
sc.set_search_model(w.Qs_xyz)
sc.set_ppf_map(keys, counts, pairs)
rng = np.random.default_rng(3)
ids, inv, status = sc.select_bases(rng.random((256, 4)))
ids, inv = ids[status == 1][:100], inv[status == 1][:100]
base_xyz = w.P_xyz[ids]
grp = MultiGpuScorer(list(range(n_dev)))
grp.init_object(0, w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
grp.set_object_search_model(0, w.Qs_xyz)
grp.set_object_ppf_map(0, keys, counts, pairs)
nq = sc.find_congruent_batch(ids, base_xyz, inv, w.delta)
picks = np.array([(b, j) for b in range(len(nq)) for j in range(min(int(nq[b]), 100))], np.int32).reshape(-1, 2)

def t(fn, n=20):
    fn(); fn()
    gc.collect(); gc.disable()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    gc.enable()
    return f"median {np.median(ts):.3f} min {min(ts):.3f} max {max(ts):.3f} ms"

print("single find_congruent_batch:", t(lambda: sc.find_congruent_batch(ids, base_xyz, inv, w.delta)))
print("group  find_congruent_batch:", t(lambda: grp.find_congruent_batch(0, ids, base_xyz, inv, w.delta)))
print("single congruent_batch_fit :", t(lambda: sc.congruent_batch_fit(picks, ids, w.centroid_P, w.centroid_Q)))
print("group  congruent_batch_fit :", t(lambda: grp.congruent_batch_fit(0, picks, ids, w.centroid_P, w.centroid_Q)))
print("single quads               :", t(lambda: sc.congruent_batch_quads(picks)))
print("group  quads               :", t(lambda: grp.congruent_batch_quads(0, picks)))
