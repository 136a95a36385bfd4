"""Depth decode + back-projection (PPE/misc/utilities.cpp:47-61, 190-206 and
Segmentation.cpp:219): the C restatement and the HIP kernel against the committed fixture produced by
the Eigen-typed harness from a window of the reference's own test-scene/ frame.  Bit-exact: the
expected clouds are stored as count + SHA-256 of their float32 bytes + every 97th point."""
import hashlib
import os

import numpy as np
import pytest

from _checkers import have_ref, oracle_backproject

GOLD = os.path.join(os.path.dirname(__file__), "golden", "backproject.npz")


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a, np.float32).tobytes()).digest(), np.uint8)


def check(fn):
    g = np.load(GOLD)
    raw, mask, K = g["raw"], g["mask"], g["K"]
    cloud = fn(raw, None, K)
    assert len(cloud) == int(g["all_n"]) and np.array_equal(cloud[::97], g["all_sample"])
    assert np.array_equal(sha(cloud), g["all_sha"])
    for cls in (2, 3, 8):
        cloud = fn(raw, (mask == cls).astype(np.uint8), K)
        assert len(cloud) == int(g[f"n_{cls}"])
        assert np.array_equal(cloud[::97], g[f"sample_{cls}"]) and np.array_equal(sha(cloud), g[f"sha_{cls}"])
    # the float-image form: decode here as utilities.cpp:57-59 does, same cloud
    depth = ((((raw.astype(np.uint32) << 13) | (raw >> 3)) & 0xFFFF).astype(np.float32) / np.float32(10000))
    assert np.array_equal(sha(depth), g["depth_sha"])
    assert np.array_equal(sha(fn(depth, None, K)), g["all_sha"])
    # scan order and the range rule
    c = fn(depth, None, K)
    z = depth[(depth > 0.1) & (depth < 2.0)]
    assert np.array_equal(c[:, 2], z)


def test_oracle_matches_fixture():
    check(oracle_backproject)


def test_oracle_edges():
    K = np.array([[500, 0, 2.5], [0, 400, 1.5], [0, 0, 1]], np.float32)
    d = np.array([[0.0, 0.1, 0.10000001, 1.0], [1.9999999, 2.0, np.nan, np.inf], [-1.0, 0.5, 0.5, 0.5]], np.float32)
    c = oracle_backproject(d, None, K)
    keep = (d.astype(np.float64) > 0.1) & (d.astype(np.float64) < 2.0)
    assert len(c) == keep.sum() == 7      # 0.1f is ABOVE the double literal 0.1: the reference keeps it
    m = np.ones_like(d, np.uint8)
    m[2, 1] = 0
    assert len(oracle_backproject(d, m, K)) == 6
    assert len(oracle_backproject(d[:0], None, K)) == 0
    assert len(oracle_backproject(np.zeros((3, 4), np.uint16), None, K)) == 0


@pytest.mark.skipif(not have_ref(), reason="needs oracle/_ref (build container)")
def test_oracle_vs_harness_random_images():
    from _checkers import ref_backproject
    rng = np.random.default_rng(4)
    for rows, cols in ((1, 1), (7, 13), (64, 96), (480, 640)):
        raw = rng.integers(0, 65536, (rows, cols)).astype(np.uint16)
        mask = (rng.random((rows, cols)) < 0.7).astype(np.uint8)
        K = np.array([[rng.uniform(300, 700), 0, cols / 2 + rng.normal()], [0, rng.uniform(300, 700), rows / 2 + rng.normal()],
                      [0, 0, 1]], np.float32)
        _, c_ref = ref_backproject(raw, mask, K)
        assert np.array_equal(oracle_backproject(raw, mask, K), c_ref)


@pytest.mark.gpu
def test_hip_matches_fixture_and_oracle():
    from physimglobalpose_amd import LcpScorer
    sc = LcpScorer(0)
    check(lambda img, m, K: sc.backproject_depth(img, K, m))
    rng = np.random.default_rng(8)
    for rows, cols in ((1, 1), (3, 5), (255, 257), (480, 640), (1080, 1920)):
        raw = rng.integers(0, 65536, (rows, cols)).astype(np.uint16)
        mask = (rng.random((rows, cols)) < 0.6).astype(np.uint8)
        K = np.array([[610.0, 0, cols / 2 + 0.37], [0, 612.5, rows / 2 - 0.21], [0, 0, 1]], np.float32)
        assert np.array_equal(sc.backproject_depth(raw, K, mask), oracle_backproject(raw, mask, K))
        assert np.array_equal(sc.backproject_depth(raw, K), oracle_backproject(raw, None, K))
    d = np.array([[0.0, 0.1, 0.10000001, 1.0], [1.9999999, 2.0, np.nan, np.inf]], np.float32)
    K = np.array([[500, 0, 2.5], [0, 400, 1.5], [0, 0, 1]], np.float32)
    assert np.array_equal(sc.backproject_depth(d, K), oracle_backproject(d, None, K))
    assert len(sc.backproject_depth(d[:0], K)) == 0
    assert len(sc.backproject_depth(d, K, z_min=0.5, z_max=1.5)) == 1


@pytest.mark.gpu
def test_hip_segment_feeds_the_scorer():
    """image -> cloud -> radius filter -> scene index: the chain in front of the path on one context."""
    from physimglobalpose_amd import LcpScorer
    g = np.load(GOLD)
    sc = LcpScorer(0)
    cloud = sc.backproject_depth(g["raw"], g["K"], (g["mask"] == 8).astype(np.uint8))
    keep, _ = sc.radius_outlier_filter(cloud, None, 0.03, 10)
    assert keep.sum() > 0.9 * len(cloud)
