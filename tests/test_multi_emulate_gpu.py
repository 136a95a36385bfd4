"""Every N > 1 branch of the native multi-GPU group (csrc/multi_gpu.hip) on ONE device:
PGP_MULTI_EMULATE=n makes n logical members (own context, worker thread and stream each) share the
GPU, the exchange being a sum kernel with the all-reduce's semantics (RCCL refuses one device twice).
Slices, zeroed full-length vectors, the exchange, the arg-max with near-tie settlement and the exact
running-best records across slice boundaries must give what a single context gives, bit for bit
(SceneCfg.cpp:376-406 / HypothesisSelection.cpp:248-257 read these arrays)."""
import os

import numpy as np
import pytest

from physimglobalpose_amd import LcpScorer, MultiGpuScorer, PGP_MODE_PLAIN, PGP_MODE_WEIGHTED, synth
from _checkers import Oracle

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _group(monkeypatch, n):
    monkeypatch.setenv("PGP_MULTI_EMULATE", str(n))
    grp = MultiGpuScorer([0])
    assert grp.n_devices == n
    return grp


@pytest.mark.parametrize("n", [2, 3, 8])
def test_emulated_group_equals_single_context(n, monkeypatch):
    w = synth.make_workload(20000, 2000, 777, config_id=41)
    one = LcpScorer(0)
    one.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    grp = _group(monkeypatch, n)
    grp.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    for mode in (PGP_MODE_PLAIN, PGP_MODE_WEIGHTED):
        for m in (777, 100, n, n - 1, 1, 0):          # also fewer hypotheses than members: empty slices
            a = one.score(w.T[:m], mode, w.gate_deg)
            b = grp.score(w.T[:m], mode, w.gate_deg)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2:] == b[2:], (mode, m)
        grp.upload(w.T[:300])
        c = grp.score_uploaded(mode, w.gate_deg)
        d = one.score(w.T[:300], mode, w.gate_deg)
        assert np.array_equal(c[0], d[0]) and np.array_equal(c[1], d[1]) and c[2:] == d[2:]
    grp.close()


@pytest.mark.parametrize("n", [2, 3, 8])
def test_near_tie_cluster_straddling_slice_boundaries(n, monkeypatch):
    g = np.load(os.path.join(GOLD, "near_ties.npz"))
    n_h = len(g["T"])
    owners = {next(k for k in range(n) if MultiGpuScorer.slice_of(n_h, k, n)[0] <= i < MultiGpuScorer.slice_of(n_h, k, n)[1])
              for i in g["cluster"]}
    assert len(owners) >= 2          # the near-tie cluster lives on several members
    grp = _group(monkeypatch, n)
    grp.init(g["P"], g["Pn"], g["Pw"], g["Q"], g["Qn"], float(g["delta"]))
    s, c, bi, bs = grp.score(g["T"], PGP_MODE_WEIGHTED, 30.0)
    assert bi == int(g["best_weighted"]) and np.float32(bs) == g["wscores"][bi]
    assert np.allclose(s, g["wscores"], rtol=0, atol=2e-6)
    sp, cp, bip, _ = grp.score(g["T"], PGP_MODE_PLAIN)
    assert np.array_equal(cp, g["counts"]) and bip == int(g["best_plain"])
    grp.close()


@pytest.mark.parametrize("n", [2, 8])
def test_exact_records_across_slices(n, monkeypatch):
    """the running-best list of the COMPLETE vector is the reference's, although its near-records sit on different members"""
    w = synth.make_workload(20000, 2000, 1024, config_id=3)
    rng = np.random.default_rng(7)
    one = LcpScorer(0)
    one.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    s0, _, bi0, _ = one.score(w.T, PGP_MODE_WEIGHTED, w.gate_deg)
    base = w.T[bi0].reshape(4, 4, order="F").astype(np.float64)
    crowd = np.stack([synth.colmajor16(synth._se3(synth._random_rot(rng, 2e-4), 2e-5 * rng.standard_normal(3)) @ base)
                      for _ in range(1500)])
    T = np.concatenate([w.T, crowd])[rng.permutation(1024 + 1500)]      # near-records everywhere in the batch
    one.set_exact_records(True)
    a = one.score(T, PGP_MODE_WEIGHTED, w.gate_deg)
    grp = _group(monkeypatch, n)
    grp.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    grp.set_exact_records(True)
    b = grp.score(T, PGP_MODE_WEIGHTED, w.gate_deg)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2:] == b[2:]
    orc = Oracle(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm)
    so, bio, selo = orc.score_batch(T, w.delta, mode=1, gate_deg=w.gate_deg, threads=8)
    assert b[2] == bio and np.array_equal(LcpScorer.running_best(b[0]), selo) and np.array_equal(b[0][selo], so[selo])
    grp.close()
