"""Parity on the benchmark's OWN workload, in full (bench.py's batches were oracle-checked on a 65-hypothesis
sample only).  bench.py times synth.make_workload(50 000, 5 000, 8 x 4096, config_id=2) (BASELINE.json configs[1], eight
distinct batches in rotation) and, for the configs[2] row, make_workload(50 000, 5 000, 16 384, config_id=210).  Every
hypothesis of both is scored by the C oracle (OpenMP, the restatement of base.cc:1699-1766 pinned on the reference's kd-tree)
and compared with the HIP path through the C ABI:
  plain    Verify          : scores bit-exact, best index per batch identical
  weighted WeightedVerify  : scores within 2e-6 (bar of north_star: 1e-4), best index per batch identical under the
                             reference's strict `>` walk (base.cc:1891-1901), and the
                             registered scene ids of each batch's winner identical."""
import os

import numpy as np
import pytest

from physimglobalpose_amd import LcpScorer, PGP_MODE_PLAIN, PGP_MODE_WEIGHTED, synth
from _checkers import Oracle

pytestmark = pytest.mark.gpu
THREADS = max(1, min(16, os.cpu_count() or 1))


def _full_parity(w, batch):
    sc = LcpScorer()
    sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    orc = Oracle(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm)
    n = len(w.T)
    assert n % batch == 0
    worst = 0.0
    for b in range(n // batch):
        T = w.T[b * batch:(b + 1) * batch]
        # plain
        s, c, bi, bs = sc.score(T, PGP_MODE_PLAIN)
        so, bio, _ = orc.score_batch(T, w.delta, mode=0, threads=THREADS)
        assert np.array_equal(s, so), (b, int((s != so).sum()))
        assert bi == bio and bs == so[bio]
        assert np.array_equal(c, np.rint(so.astype(np.float64) * len(w.Q_xyz)).astype(np.int32))
        # weighted
        sw, cw, biw, bsw = sc.score(T, PGP_MODE_WEIGHTED, w.gate_deg)
        swo, biwo, _ = orc.score_batch(T, w.delta, mode=1, gate_deg=w.gate_deg, threads=THREADS)
        err = float(np.abs(sw.astype(np.float64) - swo).max())
        worst = max(worst, err)
        assert err <= 2e-6, (b, err)
        assert biw == biwo, (b, biw, biwo, sw[biw], swo[biwo])
        assert abs(float(bsw) - float(swo[biwo])) <= 2e-6   # (bit-exact when a runner-up forced the settlement)
        ws, reg = orc.weighted_verify(T[biw], w.delta, w.gate_deg)
        assert np.array_equal(sc.registered(T[biw], PGP_MODE_WEIGHTED, w.gate_deg), reg) and cw[biw] == len(reg)
    return worst


def test_every_hypothesis_of_the_bench_batches():
    """8 x 4096 hypotheses = what `python bench.py` rotates through at N = 1."""
    w = synth.make_workload(50000, 5000, 4096 * 8, config_id=2)
    worst = _full_parity(w, 4096)
    print(f"weighted: largest |gpu - oracle| over 32 768 hypotheses = {worst:.3e}")


def test_every_hypothesis_of_the_config2_object_batch():
    """the 16 384-hypothesis batch of bench.py's config2_object row."""
    w = synth.make_workload(50000, 5000, 16384, config_id=210)
    _full_parity(w, 16384)
