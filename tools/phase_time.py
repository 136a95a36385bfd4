#!/usr/bin/env python3
"""Where a wave of the scoring kernel spends its time: run the PGP_ABLATE=10 build (s_memtime stamps around
the phases of a wave-iteration, `make -C physimglobalpose_amd/csrc ablate N=10`) on the C2 batch and print the
per-phase share of all wave-cycles.  The stamps wait for scalar loads, so this is a decomposition of a slightly
disturbed kernel, not the kernel's speed.  GPU box:
    cp tools/ab/libpgp_ablate10.so physimglobalpose_amd/libpgp.so && python tools/phase_time.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from physimglobalpose_amd import LcpScorer, synth, _lib, PGP_MODE_PLAIN, PGP_MODE_WEIGHTED  # noqa: E402

lib = C.CDLL(_lib.LIB_PATH)
lib.pgp_debug_phase_cycles.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
w = synth.make_workload(50000, 5000, 4096, config_id=2)
sc = LcpScorer(0)
sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
sc.reserve(4096)
dT = torch.from_numpy(w.T).cuda()
ds = torch.zeros(4096, device="cuda")
names = ["transform + cell + occupancy word (every trip)", "run descriptors + slot scan", "owner table to LDS",
         "candidate batches (owner resolution, gathers, tests, result atomics)", "result read-back + gate + weight",
         "non-empty part of a trip, total"]
for mode, mname in ((PGP_MODE_PLAIN, "plain"), (PGP_MODE_WEIGHTED, "weighted")):
    buf = (C.c_ulonglong * 16)()
    sc.score_device(dT, ds, mode=mode)
    torch.cuda.synchronize()
    lib.pgp_debug_phase_cycles(buf, 1)
    sc.score_device(dT, ds, mode=mode)   # the rows hold the last launch
    torch.cuda.synchronize()
    lib.pgp_debug_phase_cycles(buf, 1)
    v = np.array(list(buf), dtype=np.float64)
    waves = v[7]
    trips = 4096 * 20 * 4 / waves
    print(f"{mname}: {waves:.0f} waves in the launch, {trips:.1f} trips each; s_memtime ticks per TRIP:")
    for k, nm in enumerate(names):
        print(f"   {v[k] / waves / trips:10.1f}   {nm}")
