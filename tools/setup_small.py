import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, tempfile
from physimglobalpose_amd import LcpScorer
from _dropin import make_dropin_case
with tempfile.TemporaryDirectory() as d:
    _, case = make_dropin_case(d)
w = case["w"]
sc = LcpScorer()
def t(fn, n=50):
    fn(); t0=time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter()-t0)/n*1e3
print("n_scene", len(w.P_xyz), "n_model", len(w.Q_xyz), "n_search", len(w.Qs_xyz))
print("set_scene %.3f ms" % t(lambda: sc.set_scene(w.P_xyz, w.P_nrm, None, w.delta)))
print("set_model %.3f ms" % t(lambda: sc.set_model(w.Q_xyz, w.Q_nrm)))
print("set_search_model %.3f ms" % t(lambda: sc.set_search_model(w.Qs_xyz)))
print("set_scene_weights %.3f ms" % t(lambda: sc.set_scene_weights(w.P_w)) if hasattr(sc,'set_scene_weights') else '')
