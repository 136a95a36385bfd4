"""bench.py's config2_three_objects row alone (quick): python tools/three_objects.py"""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from physimglobalpose_amd import synth

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, r

w = synth.make_workload(50000, 5000, 16384, config_id=210)
print(json.dumps(bench.three_objects_row(torch, timed, w), indent=1))

