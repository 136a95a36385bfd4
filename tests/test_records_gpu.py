"""The running-best LIST in weighted mode (base.cc:1891-1908 = the drop-in's hypothesisSet): with
pgp_set_exact_records the scores at every decision of the walk are the reference's sequential sums, so
the list equals the oracle's entry for entry -- also on batches crowded with near-equal hypotheses."""
import numpy as np
import pytest

from physimglobalpose_amd import LcpScorer, PGP_MODE_WEIGHTED, PGP_MODE_PLAIN, synth
from _checkers import Oracle

pytestmark = pytest.mark.gpu


def crowded_batch(w, rng, n):
    """hypotheses within microns of each other: different inlier sets with nearly equal weighted sums"""
    base = w.T[int(np.argmax([0]))].reshape(4, 4, order="F").astype(np.float64)
    out = []
    for _ in range(n):
        d = synth._se3(synth._random_rot(rng, 2e-4), 2e-5 * rng.standard_normal(3))
        out.append(synth.colmajor16(d @ base))
    return np.stack(out)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_running_best_list_equals_the_reference_walk(seed):
    w = synth.make_workload(20000, 2000, 1024, config_id=2 + seed)
    rng = np.random.default_rng(100 + seed)
    sc = LcpScorer(0)
    sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    orc = Oracle(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm)
    # the mixed batch (its records accumulate), then its best pose's near-copies, some a hair better
    s0, _, bi0, _ = sc.score(w.T, PGP_MODE_WEIGHTED, w.gate_deg)
    wb = type("W", (), {"T": w.T[bi0:bi0 + 1]})
    T = np.concatenate([w.T, crowded_batch(wb, rng, 2000)])
    sc.set_exact_records(True)
    s, c, bi, bs = sc.score(T, PGP_MODE_WEIGHTED, w.gate_deg)
    so, bio, selo = orc.score_batch(T, w.delta, mode=1, gate_deg=w.gate_deg, threads=8)
    assert bi == bio and bs == so[bio]
    sel = LcpScorer.running_best(s)
    assert np.array_equal(sel, selo)
    assert np.array_equal(s[selo], so[selo])              # the records carry the reference's bits
    assert np.allclose(s, so, rtol=0, atol=2e-6)           # everything else as before
    assert len(selo) >= 3
    # plain mode is untouched by the option (counts are exact anyway)
    s, _, bi, _ = sc.score(T, PGP_MODE_PLAIN)
    so, bio, selo = orc.score_batch(T, w.delta, mode=0, threads=8)
    assert np.array_equal(s, so) and bi == bio and np.array_equal(LcpScorer.running_best(s), selo)


def test_option_off_leaves_the_scores_alone():
    w = synth.make_workload(8000, 800, 256, config_id=2)
    sc = LcpScorer(0)
    sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    a, _, _, _ = sc.score(w.T, PGP_MODE_WEIGHTED, w.gate_deg)
    sc.set_exact_records(True)
    b, _, _, _ = sc.score(w.T, PGP_MODE_WEIGHTED, w.gate_deg)
    sc.set_exact_records(False)
    c, _, _, _ = sc.score(w.T, PGP_MODE_WEIGHTED, w.gate_deg)
    assert np.array_equal(a, c)
    assert np.allclose(a, b, rtol=0, atol=2e-6) and (a != b).sum() <= 64   # only near-records were re-scored
