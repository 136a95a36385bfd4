"""The native multi-GPU group and `bench.py --gpus N` over PHYSICAL devices and real RCCL (north_star: "partition the
hypothesis batch across the 8 GPUs of one node with an RCCL all-reduce over xGMI of the per-hypothesis LCP scores";
consumers SceneCfg.cpp:376-406, HypothesisSelection.cpp:248-257).  The 1-GPU boxes of the build pool cannot run this:
every test here is skipped unless torch.cuda.device_count() >= 2, so the first multi-GPU box that runs
`pytest -m gpu` proves -- or refutes -- the path without another round:
  (i)  MultiGpuScorer(range(n)) == a single context, bit for bit: both modes, batches shorter than the group,
       resident transforms, the weighted near-tie cluster of near_ties.npz straddling a slice boundary, the exact
       running-best records across slices;
  (ii) `torch.distributed.run --nproc-per-node n bench.py --gpus n` under the nccl (= RCCL) backend prints the ONE
       JSON line with the per-call form and the native group's row, none of them carrying an error."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from physimglobalpose_amd import LcpScorer, MultiGpuScorer, PGP_MODE_PLAIN, PGP_MODE_WEIGHTED, synth
from _checkers import Oracle
from test_multi_objects_gpu import congruent_case  # noqa: F401  (module-scoped fixture)

N_DEV = torch.cuda.device_count()      # counting devices does not initialise the GPU
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(N_DEV < 2, reason=f"needs >= 2 physical GPUs (this box has {N_DEV})")]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
GROUPS = sorted({2, min(N_DEV, 4), N_DEV} - {0, 1}) if N_DEV >= 2 else [2]


@pytest.fixture(autouse=True)
def _real_devices(monkeypatch):
    monkeypatch.delenv("PGP_MULTI_EMULATE", raising=False)      # physical members, the real collective
    monkeypatch.delenv("PGP_MULTI_FORCE_COLLECTIVE", raising=False)


@pytest.mark.parametrize("n", GROUPS)
def test_group_over_physical_devices_equals_single_context(n):
    w = synth.make_workload(20000, 2000, 4096, config_id=41)
    one = LcpScorer(0)
    one.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    grp = MultiGpuScorer(list(range(n)))
    assert grp.n_devices == n
    grp.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    for mode in (PGP_MODE_PLAIN, PGP_MODE_WEIGHTED):
        for m in (4096, 777, n, n - 1, 1, 0):          # also fewer hypotheses than members: empty slices
            a = one.score(w.T[:m], mode, w.gate_deg)
            b = grp.score(w.T[:m], mode, w.gate_deg)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2:] == b[2:], (mode, m)
        grp.upload(w.T[:300])
        c = grp.score_uploaded(mode, w.gate_deg)
        d = one.score(w.T[:300], mode, w.gate_deg)
        assert np.array_equal(c[0], d[0]) and np.array_equal(c[1], d[1]) and c[2:] == d[2:]
    # and the oracle on the group's weighted winner
    orc = Oracle(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm)
    s, c, bi, bs = grp.score(w.T, PGP_MODE_WEIGHTED, w.gate_deg)
    so, bio, _ = orc.score_batch(w.T, w.delta, mode=1, gate_deg=w.gate_deg, threads=8)
    assert bi == bio and np.allclose(s, so, rtol=0, atol=2e-6)
    grp.close()


@pytest.mark.parametrize("n", GROUPS)
def test_near_tie_cluster_straddling_physical_slices(n):
    g = np.load(os.path.join(GOLD, "near_ties.npz"))
    n_h = len(g["T"])
    owners = {next(k for k in range(n) if MultiGpuScorer.slice_of(n_h, k, n)[0] <= i < MultiGpuScorer.slice_of(n_h, k, n)[1])
              for i in g["cluster"]}
    assert len(owners) >= 2          # the near-tie cluster lives on several devices
    grp = MultiGpuScorer(list(range(n)))
    grp.init(g["P"], g["Pn"], g["Pw"], g["Q"], g["Qn"], float(g["delta"]))
    s, c, bi, bs = grp.score(g["T"], PGP_MODE_WEIGHTED, 30.0)
    assert bi == int(g["best_weighted"]) and np.float32(bs) == g["wscores"][bi]
    assert np.allclose(s, g["wscores"], rtol=0, atol=2e-6)
    sp, cp, bip, _ = grp.score(g["T"], PGP_MODE_PLAIN)
    assert np.array_equal(cp, g["counts"]) and bip == int(g["best_plain"])
    grp.close()


def test_exact_records_across_physical_slices():
    n = GROUPS[-1]
    w = synth.make_workload(20000, 2000, 1024, config_id=3)
    rng = np.random.default_rng(7)
    one = LcpScorer(0)
    one.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    s0, _, bi0, _ = one.score(w.T, PGP_MODE_WEIGHTED, w.gate_deg)
    base = w.T[bi0].reshape(4, 4, order="F").astype(np.float64)
    crowd = np.stack([synth.colmajor16(synth._se3(synth._random_rot(rng, 2e-4), 2e-5 * rng.standard_normal(3)) @ base)
                      for _ in range(1500)])
    T = np.concatenate([w.T, crowd])[rng.permutation(1024 + 1500)]
    one.set_exact_records(True)
    a = one.score(T, PGP_MODE_WEIGHTED, w.gate_deg)
    grp = MultiGpuScorer(list(range(n)))
    grp.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    grp.set_exact_records(True)
    b = grp.score(T, PGP_MODE_WEIGHTED, w.gate_deg)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2:] == b[2:]
    grp.close()


@pytest.mark.parametrize("n", GROUPS)
def test_objects_icp_shards_and_congruent_shards_over_physical_devices(n, congruent_case):
    """row e-2 (SURVEY 8e line 4, configs[3]) over physical members: the cases of tests/test_multi_objects_gpu.py with
    the real all-reduce -- several objects in one group, ICP pose shards, congruent sets sharded by base"""
    from test_multi_objects_gpu import (check_congruent_shards_equal_single_context, check_icp_shards_equal_single_calls,
                                        check_objects_equal_single_contexts)
    shapes = [(9000, 1500, 700), (4000, 800, 0), (12000, 2000, 333), (3000, 500, 2), (6000, 1000, 1201)]
    objs = [synth.make_workload(p, q, max(h, 1), config_id=500 + k) for k, (p, q, h) in enumerate(shapes)]
    grp = MultiGpuScorer(list(range(n)))
    for _ in range(len(objs) - 1):
        grp.add_object()
    check_objects_equal_single_contexts(grp, objs, [h for _, _, h in shapes])
    check_icp_shards_equal_single_calls(grp)
    check_congruent_shards_equal_single_context(grp, congruent_case, obj=1)
    grp.close()


def test_six_objects_64k_hypotheses_over_physical_devices():
    """BASELINE.json configs[3] as the reference states it: 6 objects, 65 536 hypotheses, every visible device"""
    from test_multi_objects_gpu import check_objects_equal_single_contexts
    counts = [16384, 12288, 12288, 8192, 8192, 8192]
    objs = [synth.make_workload(20000, 3000, c, config_id=300 + k) for k, c in enumerate(counts)]
    grp = MultiGpuScorer(list(range(N_DEV)))
    for _ in range(5):
        grp.add_object()
    check_objects_equal_single_contexts(grp, objs, counts, modes=(PGP_MODE_WEIGHTED,))
    grp.close()


@pytest.mark.parametrize("n", GROUPS[:2])
def test_bench_under_real_rccl(n):
    """one process per GPU, backend "nccl" (RCCL over xGMI), the driver's own launch line"""
    env = {k: v for k, v in os.environ.items() if k not in ("PGP_DIST_BACKEND", "PGP_MULTI_EMULATE")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
                          "--master-addr", "127.0.0.1", "--master-port", str(29560 + n), "bench.py", "--gpus", str(n),
                          "--steps", "6", "--warmup", "2"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = out.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout[-500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["scaling"] == "weak" and d["value"] > 1e6
    # the headline went through the PRODUCT's group: one member per rank, ncclCommInitRank, n ranks in the communicator
    assert "native_group_error" not in d, d.get("native_group_error")
    assert d["rccl_ranks"] == n and d["devices"] == [0] and d["equals_single_device"] is True and d["exchanges_issued"] > 6
    pc = d["per_call"]
    assert pc["host_pointers"]["median_ms"] > 0 and pc["resident"]["median_ms"] > 0
    tw = d["rows"]["torch_twin"]                                   # the Python twin, secondary
    assert tw["value"] > 1e6 and tw["ms_per_step"] > 0
    nm = d["rows"].get("native_multi")
    assert nm is not None and "error" not in nm, nm
    assert nm["devices"] == n and nm["equals_single_device"] is True
    for k in ("objects", "icp_shards", "congruent_shards"):      # row e-2 over the physical devices
        assert nm[k]["equals_single_context"] is True and nm[k]["ms_per_call"] > 0, k


@pytest.mark.parametrize("n", GROUPS[:2])
def test_bench_plain_python_launch(n):
    """`python bench.py --gpus n` with NO launcher: one child process drives libpgp's single-process group over n physical
    devices (ncclCommInitAll), a second child the torch twin; rc 0, ONE line, rccl_ranks = n."""
    env = {k: v for k, v in os.environ.items() if k not in ("PGP_DIST_BACKEND", "PGP_MULTI_EMULATE", "WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    out = subprocess.run([sys.executable, "bench.py", "--gpus", str(n), "--steps", "6", "--warmup", "2"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    lines = out.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout[-500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["scaling"] == "weak" and d["value"] > 1e6 and "emulated" not in d
    assert d["rccl_ranks"] == n and d["devices"] == list(range(n)) and d["equals_single_device"] is True
    assert d["per_call"]["host_pointers"]["median_ms"] > 0
    assert "error" not in d["rows"]["torch_twin"], d["rows"]["torch_twin"]


@pytest.mark.parametrize("n", GROUPS)
def test_streaming_form_over_physical_devices(n):
    """pgp_multi_enqueue_slot / _collect with the real all-reduce on the members' second streams"""
    from test_multi_streaming_gpu import _check_streaming, _workload
    w, lists = _workload()
    one = LcpScorer(0)
    one.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    grp = MultiGpuScorer(list(range(n)))
    assert grp.info()["rccl_ranks"] == n
    _check_streaming(grp, one, w, lists)
    grp.close()
