"""A device group that spans PROCESSES (pgp_multi_create_ranked: what every rank of `torch.distributed.run ... bench.py --gpus N`
holds) with 2 and 3 ranks on this box's ONE device: PGP_MULTI_EMULATE_RANKED replaces the RCCL all-reduce (which refuses one
device twice) by an exchange through shared memory -- slices by global rank, every process holding all transforms and taking the
arg-max (near-tie settlement, exact records) over the complete vector are the production code.  Every rank compares the
synchronous call and the streaming form with a single context of its own, bit for bit (tools/ranked_member.py).
Consumers: SceneCfg.cpp:376-406, HypothesisSelection.cpp:248-257 in a node started once per GPU."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 3])
def test_ranked_group_of_several_processes_on_one_device(world):
    env = {k: v for k, v in os.environ.items() if k not in ("PGP_MULTI_EMULATE", "PGP_MULTI_FORCE_COLLECTIVE")}
    env["PGP_MULTI_EMULATE_RANKED"] = "1"
    uid = os.urandom(128).hex()
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "ranked_member.py"), str(r), str(world), uid], cwd=ROOT, env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=600) for p in procs]
    for r, (p, (out, err)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, (r, out[-500:], err[-2000:])
        assert f"RANK_OK {r} of {world}" in out and "emulated True" in out, out[-500:]
