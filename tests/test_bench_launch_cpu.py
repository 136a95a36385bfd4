"""`python bench.py --gpus N` without a launcher must never raise: the parent process stays off the GPU and
starts its measurements as children.  On a machine WITHOUT a HIP device (this build container) both children fail -- the device
group with libpgp's "no HIP device ... no CPU fallback", the torch ranks with theirs -- and the parent still prints ONE JSON
line that says so, with exit code 1 (no headline exists; a headline is never invented)."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.device_count() > 0, reason="a GPU is present: tests/test_bench_gpu.py covers the launcher-less run")
def test_launcherless_run_without_a_device_reports_instead_of_raising():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PGP_MULTI_EMULATE")}
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=600)
    lines = out.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), (out.stdout[-500:], out.stderr[-500:])
    d = json.loads(lines[0])
    assert out.returncode == 1 and d["value"] is None and d["n_gpus"] == 2
    assert "no HIP device" in d["error"]["device_group"] and "no CPU fallback" in d["error"]["device_group"]
    assert d["error"]["torch_ranks"]
    assert "Traceback" not in out.stderr        # the parent itself raised nothing
