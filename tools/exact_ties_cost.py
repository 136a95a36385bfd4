#!/usr/bin/env python3
"""What pgp_set_exact_ties costs at C2: scene set-up (the reference's kd-tree is built on the host) and the
weighted scoring step (tie flags on the candidate atomics, the tree asked for tied candidates)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from physimglobalpose_amd import LcpScorer, PGP_MODE_WEIGHTED, synth  # noqa: E402

w = synth.make_workload(50000, 5000, 4096, config_id=2)
for on in (False, True):
    sc = LcpScorer()
    sc.set_exact_ties(on)
    sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    t0 = time.perf_counter()
    for _ in range(5):
        sc.set_scene(w.P_xyz, w.P_nrm, w.P_w, w.delta)
    t_scene = (time.perf_counter() - t0) / 5
    sc.reserve(4096)
    dT = torch.from_numpy(w.T).cuda()
    ds = torch.zeros(4096, device="cuda")
    for _ in range(20):
        sc.score_device(dT, ds, mode=PGP_MODE_WEIGHTED)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        sc.score_device(dT, ds, mode=PGP_MODE_WEIGHTED)
    torch.cuda.synchronize()
    t_step = (time.perf_counter() - t0) / 200
    print(f"exact ties {'on ' if on else 'off'}: pgp_set_scene {t_scene * 1e3:.2f} ms   weighted step {t_step * 1e6:.1f} us", flush=True)
