"""The numpy restatement of the device depth renderer and of UCTState::computeCost
(oracle/render_oracle.py) on cases small enough to check by hand -- the checker of tests/test_render_gpu.py
must itself be right (UCTState.cpp:44-72,93-116; renderScene.cpp:45-72)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import render_oracle as ro  # noqa: E402

CAM = dict(rows=8, cols=10, fx=10.0, fy=10.0, cx=5.0, cy=4.0, z_near=0.1, z_max=1.0)
I16 = np.eye(4, dtype=np.float32).T.reshape(16)        # identity, column-major


def test_splat_puts_each_point_in_its_nearest_pixel_and_keeps_the_nearest_depth():
    pts = np.array([[0.0, 0.0, 0.5],        # centre: u = 5, v = 4
                    [0.0, 0.0, 0.4],        # same pixel, nearer: wins
                    [0.1, -0.1, 0.5],       # u = 10*0.1/0.5+5 = 7, v = 2
                    [0.0, 0.0, 1.5],        # beyond z_max: dropped
                    [0.0, 0.0, 0.05],       # nearer than z_near: dropped
                    [5.0, 0.0, 0.5]], np.float32)   # outside the image
    d = ro.splat(pts, I16, CAM)
    assert d[4, 5] == np.float32(0.4) and d[2, 7] == np.float32(0.5) and (d > 0).sum() == 2
    parent = np.full((8, 10), 0.45, np.float32)
    d = ro.splat(pts, I16, CAM, parent=parent)
    assert d[4, 5] == np.float32(0.4) and d[2, 7] == np.float32(0.45) and d[0, 0] == np.float32(0.45)   # UCTState.cpp:65


def test_raster_covers_the_pixel_centres_inside_a_triangle_at_its_plane_depth():
    # a fronto-parallel triangle at z = 0.5: pixel corners (2,1) (8,1) (2,7) -> centres (i + .5, j + .5) with x + y <= 9
    z = 0.5
    def world(u, v):
        return [(u - 5.0) * z / 10.0, (v - 4.0) * z / 10.0, z]
    v = np.array([world(2, 1), world(8, 1), world(2, 7)], np.float32)
    d = ro.raster(v, np.array([[0, 1, 2]]), I16, CAM)
    want = np.zeros((8, 10), bool)
    for j in range(8):
        for i in range(10):
            cx, cy = i + 0.5, j + 0.5
            want[j, i] = cx >= 2 and cy >= 1 and (cx - 2) + (cy - 1) <= 6
    assert np.array_equal(d > 0, want)
    assert np.allclose(d[d > 0], 0.5, atol=1e-6)
    # winding does not matter; a slanted triangle interpolates 1/z linearly in the image
    assert np.array_equal(ro.raster(v, np.array([[0, 2, 1]]), I16, CAM), d)
    v2 = v.copy()
    v2[1] = [(8 - 5.0) * 0.8 / 10.0, (1 - 4.0) * 0.8 / 10.0, 0.8]
    d2 = ro.raster(v2, np.array([[0, 1, 2]]), I16, CAM)
    row = d2[1, 2:8]
    assert (np.diff(row) > 0).all() and row[0] > 0.5 and row[-1] < 0.8
    inv = 1.0 / row.astype(np.float64)
    assert np.allclose(np.diff(inv), np.diff(inv)[0], rtol=1e-4)        # perspective-correct


def test_depth_cost_counts_like_compute_cost():
    obs = np.array([[0.5, 0.5, 0.0, 0.0], [0.5, 0.5, 0.5, 0.0]], np.float32)
    ren = np.array([[[0.5, 0.52, 0.3, 0.0], [0.0, 0.505, 0.5, 0.2]]], np.float32)
    # |d| > 0.01: pixels (0,1) both>0, (0,2) ren only, (1,0) obs only, (1,3) ren only
    c = ro.depth_cost(obs, ren, 0.01)
    assert c.tolist() == [[2, 3, 1]]            # obScore, renScore, intScore -> renderScore 4 (UCTState.cpp:115)


def test_a_plane_that_runs_past_the_camera_is_clipped_not_dropped():
    """a floor y = 0.2 from 1 m in front of the camera to 1 m BEHIND it (two triangles with vertices at z = -1): every
    pixel whose ray meets the floor between z_near and z_max carries the floor's depth z = 0.2 fy / (v + 0.5 - cy)"""
    cam = dict(rows=12, cols=10, fx=10.0, fy=10.0, cx=5.0, cy=4.0, z_near=0.1, z_max=1.0)
    v = np.array([[-2.0, 0.2, -1.0], [2.0, 0.2, -1.0], [2.0, 0.2, 1.0], [-2.0, 0.2, 1.0]], np.float32)
    d = ro.raster(v, np.array([[0, 1, 2], [0, 2, 3]]), I16, cam)
    for row in range(12):
        dy = row + 0.5 - 4.0
        z = 0.2 * 10.0 / dy if dy > 0 else np.inf
        if 0.1 < z <= 1.0 and z < 0.97:        # (the last row before z_max may be cut by the far edge of the quad)
            assert np.allclose(d[row, 2:8], z, rtol=2e-5), (row, d[row], z)
        elif z > 1.0 or dy <= 0:
            assert not d[row].any(), row
    assert (d > 0).sum() >= 40
    # one vertex in front only (the apex of a fan in the middle of the view); vertex order and winding do not matter
    v5 = np.concatenate([v, np.array([[0.0, 0.2, 1.0]], np.float32)])
    t1 = ro.raster(v5, np.array([[0, 1, 4]]), I16, cam)
    assert (t1 > 0).sum() > 10 and np.array_equal(t1, ro.raster(v5, np.array([[1, 4, 0]]), I16, cam))
    assert np.array_equal(t1 > 0, ro.raster(v5, np.array([[4, 1, 0]]), I16, cam) > 0)
    for row in range(6, 12):
        z = 0.2 * 10.0 / (row + 0.5 - 4.0)
        assert np.allclose(t1[row][t1[row] > 0], z, rtol=2e-5)
    # two in front
    t2 = ro.raster(v, np.array([[0, 2, 3]]), I16, cam)
    assert (t2 > 0).sum() > 10 and np.array_equal(t2, ro.raster(v, np.array([[2, 3, 0]]), I16, cam))
    # all behind: nothing; z_near = 0 clips at 1e-4 without producing NaN
    assert not ro.raster(v[[0, 1, 0]] * np.float32(1.0), np.array([[0, 1, 2]]), I16, cam).any()
    d0 = ro.raster(v, np.array([[0, 1, 2], [0, 2, 3]]), I16, dict(cam, z_near=0.0))
    assert np.isfinite(d0).all() and (d0 > 0).sum() >= (d > 0).sum()
