"""Opt-in emulation of Verify's early termination (base.cc:1699-1731; pgp_set_verify_early_out): the
plain-mode scores of hypotheses that can no longer beat the running best are the reference's
order-dependent lower bounds, bit for bit -- against the fixtures the Eigen harness wrote
(tests/golden/*.npz early_out_scores) and the C restatement on a larger batch; best index and best score
are the same with the option on and off; a device group applies it to the complete vector."""
import glob
import os

import numpy as np
import pytest

from physimglobalpose_amd import LcpScorer, MultiGpuScorer, PGP_MODE_PLAIN, PGP_MODE_WEIGHTED, synth
from _checkers import Oracle

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIXTURES = sorted(p for p in glob.glob(os.path.join(GOLD, "*.npz")) if "early_out_scores" in np.load(p).files)


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_fixture_early_out_scores(path):
    g = np.load(path)
    sc = LcpScorer(0)
    sc.init(g["P"], g["Pn"], g["Pw"], g["Q"], g["Qn"], float(g["delta"]))
    true = sc.score(g["T"], PGP_MODE_PLAIN)
    sc.set_verify_early_out(True)
    s, c, bi, bs = sc.score(g["T"], PGP_MODE_PLAIN)
    assert np.array_equal(s, g["early_out_scores"])
    assert np.array_equal(s, c.astype(np.float32) / np.float32(len(g["Q"])))
    assert (bi, bs) == true[2:] and bi == int(g["best_plain"])
    assert (s <= true[0]).all() and s[bi] == true[0][bi]
    # weighted mode has no early termination in the reference (base.cc:1733-1766): untouched by the option
    w_on = sc.score(g["T"], PGP_MODE_WEIGHTED)
    sc.set_verify_early_out(False)
    w_off = sc.score(g["T"], PGP_MODE_WEIGHTED)
    assert np.array_equal(w_on[0], w_off[0]) and w_on[2:] == w_off[2:]
    assert np.array_equal(sc.score(g["T"], PGP_MODE_PLAIN)[0], true[0])


def test_large_batch_against_the_restatement_and_in_a_group(monkeypatch):
    w = synth.make_workload(20000, 2000, 3000, config_id=61)
    orc = Oracle(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm)
    so, bio, _ = orc.score_batch(w.T, w.delta, mode=0, early_out=True, threads=1)
    sc = LcpScorer(0)
    sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    sc.set_verify_early_out(True)
    s, c, bi, bs = sc.score(w.T, PGP_MODE_PLAIN)
    assert np.array_equal(s, so) and bi == bio
    assert (s < sc.score(w.T[:1], PGP_MODE_PLAIN)[0][0] + 2).all()
    # fewer terminated hypotheses when the best comes late: order matters, as in the reference
    rev = w.T[::-1].copy()
    so_r, bio_r, _ = orc.score_batch(rev, w.delta, mode=0, early_out=True, threads=1)
    s_r, _, bi_r, _ = sc.score(rev, PGP_MODE_PLAIN)
    assert np.array_equal(s_r, so_r) and bi_r == bio_r and not np.array_equal(s_r[::-1], s)
    for n in (3,):
        monkeypatch.setenv("PGP_MULTI_EMULATE", str(n))
        grp = MultiGpuScorer([0])
        grp.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
        grp.set_verify_early_out(True)
        sg, cg, big, bsg = grp.score(w.T, PGP_MODE_PLAIN)
        assert np.array_equal(sg, so) and np.array_equal(cg, c) and big == bio
        grp.close()
