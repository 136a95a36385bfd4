"""Why are host-driven calls slower in bench.py's child process?  Runs tools/native_multi_bench.py as a child of a parent in
several states: plain; torch + HIP context alive; after an OpenMP region of the oracle's C restatement."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
def child(tag, env=None):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "native_multi_bench.py"), "--devices", "1", "--steps", "20"],
                       capture_output=True, text=True, timeout=300, env=env)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    print(tag, d["icp_shards"]["ms_per_call"], d["icp_shards"]["one_context_per_job_ms"], d["congruent_shards"]["ms_per_call"],
          d["congruent_shards"]["one_context_ms"], flush=True)
child("plain parent           ")
import torch
x = torch.zeros(1 << 20, device="cuda"); torch.cuda.synchronize()
child("parent with HIP context")
from physimglobalpose_amd import LcpScorer, synth
w = synth.make_workload(50000, 5000, 4096, config_id=2)
sc = LcpScorer(0); sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
sc.score(w.T, 1, w.gate_deg)
child("parent with a scorer   ")
import bench
try:
    bench.cpu_baseline(w, "weighted")
except Exception as e:
    print("cpu_baseline failed", e)
child("after cpu_baseline     ")
child("after cpu_baseline, OMP_WAIT_POLICY=passive in the child", dict(os.environ, OMP_WAIT_POLICY="passive"))
