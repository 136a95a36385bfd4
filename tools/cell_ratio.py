#!/usr/bin/env python3
"""Kernel time vs grid cell size (PGP_CELL_RATIO = cell edge / delta), C2, both modes."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from physimglobalpose_amd import LcpScorer, synth, PGP_MODE_PLAIN, PGP_MODE_WEIGHTED
w = synth.make_workload(50000, 5000, 4096, config_id=2)
dT = torch.from_numpy(w.T).cuda(); ds = torch.zeros(4096, device="cuda"); dc = torch.zeros(4096, dtype=torch.int32, device="cuda"); db = torch.zeros(2, dtype=torch.int32, device="cuda")
ref = None
for ratio in os.environ.get("RATIOS", "1.02,0.75,0.51,0.34").split(","):
    os.environ["PGP_CELL_RATIO"] = ratio
    sc = LcpScorer(0); sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta); sc.reserve(4096); sc.set_kernel_timing(True)
    info = sc.index_info()
    for mode, name in ((PGP_MODE_PLAIN, "plain"), (PGP_MODE_WEIGHTED, "weighted")):
        for _ in range(5): sc.score_device(dT, ds, dc, db, mode=mode)
        torch.cuda.synchronize(); sc.kernel_timing(reset=True)
        for _ in range(30): sc.score_device(dT, ds, dc, db, mode=mode)
        torch.cuda.synchronize(); n, ms = sc.kernel_timing(reset=True)
        c = dc.cpu().numpy().copy()
        if name == "plain":
            if ref is None: ref = c
            assert np.array_equal(ref, c)
        print(f"ratio {ratio:>5s} {name:8s} {ms/n*1e3:7.1f} us  cells {info['n_cells']/1e6:6.1f} M  cand {info['n_candidates']/1e6:5.2f} M  index {info['bytes_index']/1e6:6.1f} MB  build {info['build_ms']:.2f} ms")
