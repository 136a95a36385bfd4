"""pgp_select_bases alone on the drop-in's case (256 attempts per call): ms per call.  With the PGP_SEL_STOP=k variants
(make variantf FILE=base_select NAME=stopk DEFS=-DPGP_SEL_STOP=k: the kernel returns after the k-th point, wrong results) the
differences are the stages' costs."""
import sys, os, time, gc, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from physimglobalpose_amd import LcpScorer
from _dropin import make_dropin_case
gc.disable()
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
u = np.random.default_rng(3).random((256, 4))
for rep in range(3):
    sc.select_bases(u)
    t0 = time.perf_counter()
    for _ in range(50):
        ids, inv, st = sc.select_bases(u)
    print(f"select_bases, 256 attempts: {(time.perf_counter() - t0) / 50 * 1e3:.4f} ms per call ({int((st == 1).sum())} bases)", flush=True)
