"""pgp::HostOut (csrc/pgp_internal.h): small results land in a pinned area of the context before they reach the caller's
arrays.  A result larger than the area (1 MB) takes the landing buffer of its own -- the same bytes either way: the
registered ids of a 300 000-point model (1.2 MB of hits) against the count of the same hypothesis from the scoring call, and
against a second context that registers the model in two halves which do fit."""
import numpy as np
import pytest

from physimglobalpose_amd import LcpScorer, PGP_MODE_PLAIN, synth

pytestmark = pytest.mark.gpu


def test_a_result_beyond_the_landing_area_arrives_whole():
    w = synth.make_workload(20000, 3000, 4, config_id=77)
    rng = np.random.default_rng(5)
    k = rng.integers(0, len(w.Q_xyz), 300000)
    Q = (w.Q_xyz[k] + 0.0005 * rng.standard_normal((len(k), 3))).astype(np.float32)
    Qn = w.Q_nrm[k]
    big = LcpScorer(0)
    big.init(w.P_xyz, w.P_nrm, w.P_w, Q, Qn, w.delta)
    T = w.T[0]
    s, c, bi, bs = big.score(w.T[:1], PGP_MODE_PLAIN, w.gate_deg)
    ids = big.registered(T, PGP_MODE_PLAIN)
    assert len(ids) == int(c[0]) > 1000
    halves = []
    for lo, hi in ((0, 150000), (150000, 300000)):
        part = LcpScorer(0)
        part.init(w.P_xyz, w.P_nrm, w.P_w, Q[lo:hi], Qn[lo:hi], w.delta)
        halves.append(part.registered(T, PGP_MODE_PLAIN))
        part.close()
    assert np.array_equal(ids, np.concatenate(halves))
    big.close()
