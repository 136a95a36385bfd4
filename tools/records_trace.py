#!/usr/bin/env python3
"""100 weighted C2 steps with pgp_set_exact_records on, for rocprofv3 --kernel-trace --stats (per-kernel cost of the option)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from physimglobalpose_amd import LcpScorer, PGP_MODE_WEIGHTED, synth  # noqa: E402

w = synth.make_workload(50000, 5000, 4096, config_id=2)
dT = torch.from_numpy(w.T).cuda()
ds = torch.zeros(4096, device="cuda")
sc = LcpScorer()
sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
sc.reserve(4096)
sc.set_exact_records(True)
for _ in range(100):
    sc.score_device(dT, ds, mode=PGP_MODE_WEIGHTED)
torch.cuda.synchronize()
