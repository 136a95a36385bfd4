"""Does a parent process that holds a GPU context slow the native-multi child's rows down?  Runs tools/native_multi_bench.py
as a child of (a) nothing on the GPU, (b) a torch context with one tensor, (c) a torch context + a libpgp context that has
scored a batch (what bench.py holds when it starts the child); extra env for the child from the command line (K=V ...)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
extra = dict(a.split("=", 1) for a in sys.argv[1:])
cmd = [sys.executable, os.path.join(ROOT, "tools", "native_multi_bench.py"), "--devices", "1", "--mode", "weighted", "--steps", "20"]

def child(tag):
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, **extra))
    d = json.loads(r.stdout.strip().splitlines()[-1])
    c = d.get("calls_min_median_max_ms", {})
    print(f"{tag:34s} max ms: " + " ".join(f"{k}={v[2]:.2f}" for k, v in c.items()), flush=True)
    print(f"{tag:34s} congruent group {d['congruent_shards']['ms_per_call']:.3f} one {d['congruent_shards']['one_context_ms']:.3f} | "
          f"icp group {d['icp_shards']['ms_per_call']:.3f} six contexts {d['icp_shards']['one_context_per_job_ms']:.3f} | "
          f"objects group {d['objects']['ms_per_call']:.3f}", flush=True)

child("no parent context")
import torch
x = torch.zeros(1 << 20, device="cuda"); torch.cuda.synchronize()
child("parent: torch context")
import numpy as np
from physimglobalpose_amd import LcpScorer, synth
w = synth.make_workload(50000, 5000, 4096, config_id=1)
sc = LcpScorer(0)
sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
sc.score(w.T, 1)
child("parent: torch + libpgp context")
del sc
import gc; gc.collect(); torch.cuda.empty_cache()
child("parent: torch, libpgp context closed")
# (d) the parent has made a cooperative launch (the scene-sized ICP in one launch: hipLaunchCooperativeKernel)
trng = np.random.default_rng(12)
top = np.c_[trng.uniform(-0.6, 0.6, 90000), trng.uniform(-0.4, 0.4, 90000), 0.0005 * trng.standard_normal(90000)]
rim = np.c_[trng.uniform(-0.6, 0.6, 10000), np.where(trng.random(10000) < 0.5, -0.4, 0.4), trng.uniform(-0.05, 0.0, 10000)]
t_tgt = np.concatenate([top, rim]).astype(np.float32)
t_src = (t_tgt[trng.choice(len(t_tgt), 30000, replace=False)] + np.array([0.004, -0.003, 0.002])).astype(np.float32)
sc = LcpScorer(0)
sc.icp_refine_ex(t_src, t_tgt, synth.colmajor16(np.eye(4))[None], max_iterations=30, max_corr_dist=0.01, energy_ratio=0.0,
                 transformation_epsilon=1e-9, absolute_mse=1e-12)
child("parent: + one cooperative launch")
del sc
gc.collect()
child("parent: that context closed")
