"""pgp_sample_quads -- the host statement of the draw the device makes inside pgp_congruent_batch_sample_fit_score_list (the
reference's sample of at most 100 random quads per base, base.cc:1858-1866) -- is a pure host function: its contract without a
GPU.  Every base: min(n, cap) DISTINCT quads in ascending order, all of a base that has fewer than cap; a function of (seed,
base number, the base's quad count) alone; uniform over the quads."""
import numpy as np
import pytest

from physimglobalpose_amd import LcpScorer


def test_draw_contract():
    nq = np.array([0, 5, 99, 100, 101, 250, 100000, 2 ** 22 + 5, 2 ** 31 - 1], np.int32)
    p = LcpScorer.sample_quads(42, nq, 100)
    assert [int((p[:, 0] == b).sum()) for b in range(len(nq))] == [0, 5, 99, 100, 100, 100, 100, 100, 100]
    for b in range(len(nq)):
        j = p[p[:, 0] == b, 1]
        assert (np.diff(j) > 0).all() and (len(j) == 0 or (j[0] >= 0 and j[-1] < nq[b]))
    assert np.array_equal(p[p[:, 0] == 3, 1], np.arange(100))            # 100 of 100: all of them
    assert np.array_equal(p, LcpScorer.sample_quads(42, nq, 100))         # a function of its arguments
    assert not np.array_equal(p, LcpScorer.sample_quads(43, nq, 100))
    # a base's draw depends on its number and its count only
    q = LcpScorer.sample_quads(42, np.array([7, 5, 1, 100, 101], np.int32), 100)
    assert np.array_equal(p[p[:, 0] == 4, 1], q[q[:, 0] == 4, 1])


def test_draw_is_uniform():
    hits = np.zeros(1000, np.int64)
    for seed in range(400):
        hits[LcpScorer.sample_quads(seed, np.array([1000], np.int32), 100)[:, 1]] += 1
    # 40 000 draws over 1000 quads: 40 each, standard deviation ~6
    assert hits.sum() == 40000 and hits.min() >= 12 and hits.max() <= 75 and abs(hits[:500].sum() - 20000) < 600


def test_bad_arguments():
    with pytest.raises(Exception):
        LcpScorer.sample_quads(1, np.array([5], np.int32), 129)
    with pytest.raises(Exception):
        LcpScorer.sample_quads(1, np.array([-1], np.int32), 100)
