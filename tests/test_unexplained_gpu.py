"""UCTState::performTrICP's pre-filter on the device (pgp_unexplained_segment, UCTState.cpp:142-174): the
segment minus the points that already-placed objects explain, against the numpy float32 restatement
(oracle/preprocess_oracle.py), bit for bit, and against scipy's kd-tree away from the radius."""
import os
import sys

import numpy as np
import pytest

from physimglobalpose_amd import LcpScorer, synth

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import preprocess_oracle as po  # noqa: E402

pytestmark = pytest.mark.gpu


def _case(seed, n_seg=3000, sizes=(5000, 1500, 2200)):
    rng = np.random.default_rng(seed)
    models, poses, placed = [], [], []
    for k, m in enumerate(sizes):
        M = synth.make_model(rng, m)[0].astype(np.float32)
        G = synth._se3(synth._random_rot(rng), [0.25 * k - 0.2, 0.05 * k, 0.7])
        models.append(M)
        poses.append(synth.colmajor16(G))
        placed.append(M.astype(np.float64) @ G[:3, :3].T + G[:3, 3])
    placed = np.concatenate(placed)
    # a segment that overlaps the placed objects in part: noisy copies of their points + points elsewhere
    near = placed[rng.choice(len(placed), n_seg // 2)] + rng.normal(0, 0.004, (n_seg // 2, 3))
    far = rng.uniform([-0.4, -0.2, 0.5], [0.6, 0.3, 0.9], (n_seg - n_seg // 2, 3))
    seg = np.concatenate([near, far])[rng.permutation(n_seg)].astype(np.float32)
    return seg, models, np.stack(poses), placed


def test_equals_the_restatement_and_a_kd_tree():
    from scipy.spatial import cKDTree
    sc = LcpScorer()
    for seed in (1, 2):
        seg, models, poses, placed = _case(seed)
        keep = sc.unexplained_segment(seg, models, poses, 0.008)
        assert np.array_equal(keep, po.unexplained_segment(seg, models, poses, 0.008))
        d = cKDTree(placed).query(seg.astype(np.float64))[0]
        clear = np.abs(d - 0.008) > 1e-5                        # float32 vs float64 may differ only AT the radius
        assert np.array_equal(keep[clear], (d >= 0.008)[clear]) and 0.2 < keep.mean() < 0.9


def test_edge_cases():
    sc = LcpScorer()
    seg, models, poses, _ = _case(3, n_seg=700, sizes=(900, 1))
    assert sc.unexplained_segment(seg, [], np.zeros((0, 16), np.float32)).all()        # first object: nothing placed yet
    assert sc.unexplained_segment(seg[:0], models, poses).shape == (0,)
    k = sc.unexplained_segment(seg, models, poses)
    assert np.array_equal(k, po.unexplained_segment(seg, models, poses))
    # strictly below the radius: a point AT the radius stays (FLANN's radius search)
    m = np.zeros((1, 3), np.float32)
    s = np.array([[0.008, 0, 0], [0.0079999, 0, 0], [0, 0.0080001, 0]], np.float32)
    I = synth.colmajor16(np.eye(4))[None]
    want = po.unexplained_segment(s, [m], I)
    assert np.array_equal(sc.unexplained_segment(s, [m], I), want) and want.tolist() == [True, False, True]
    nan = seg.copy()
    nan[5] = np.nan
    assert sc.unexplained_segment(nan, models, poses)[5]                                  # NaN is near nothing
