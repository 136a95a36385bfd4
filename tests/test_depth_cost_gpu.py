"""MCTS leaf cost (UCTState::computeCost, PPE/hypothesis_verification/mcts/UCTState.cpp:93-116) on the
GPU vs a literal numpy restatement of the reference loop (float32 arithmetic; counts are exact)."""
import numpy as np
import pytest

from physimglobalpose_amd import LcpScorer

pytestmark = pytest.mark.gpu


def reference_cost(obs, ren, thr):
    obs, ren, thr = obs.astype(np.float32), ren.astype(np.float32), np.float32(thr)
    d = np.abs((obs - ren).astype(np.float32))
    far = d > thr
    ob = np.count_nonzero((obs > 0) & far)
    re = np.count_nonzero((ren > 0) & far)
    it = np.count_nonzero((obs > 0) & (ren > 0) & far)
    return np.float32(ob) + np.float32(re) - np.float32(it), (ob, re, it)


@pytest.mark.parametrize("shape,n", [((480, 640), 5), ((37, 53), 3), ((1, 1), 2), ((480, 641), 2)])
def test_matches_reference_loop(shape, n):
    rng = np.random.default_rng(shape[1])
    obs = rng.uniform(0.3, 1.2, shape).astype(np.float32)
    obs[rng.random(shape) < 0.3] = 0.0                     # invalid depth
    ren = np.stack([obs + rng.choice([0.0, 0.005, 0.0100001, 0.02, -0.03], shape).astype(np.float32)
                    for _ in range(n)])
    ren[rng.random(ren.shape) < 0.4] = 0.0                 # pixels the render does not cover
    ren[0] = obs                                           # identical image: cost 0
    sc = LcpScorer()
    score, counts = sc.depth_cost(obs, ren, 0.01)
    for i in range(n):
        s, c = reference_cost(obs, ren[i], 0.01)
        assert score[i] == s and tuple(counts[i]) == c
    assert score[0] == 0


def test_empty_and_nan():
    sc = LcpScorer()
    s, c = sc.depth_cost(np.zeros((4, 4), np.float32), np.zeros((0, 4, 4), np.float32))
    assert len(s) == 0
    obs = np.full((8, 8), np.nan, np.float32)
    s, c = sc.depth_cost(obs, np.ones((1, 8, 8), np.float32))
    assert s[0] == 0 and not c.any()                       # NaN compares false everywhere
