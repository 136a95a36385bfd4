"""Depth rendering of posed objects on the device (csrc/render.hip; replaces the OpenGL pass behind
UCTState::render, UCTState.cpp:44-72 / renderScene.cpp:45-72) against its numpy float32 restatement
(oracle/render_oracle.py), bit for bit, and the leaf cost computed from device-resident images
(pgp_depth_cost_device = UCTState::computeCost) against the host-pointer path on host-rendered images."""
import os
import sys

import numpy as np
import pytest

from physimglobalpose_amd import LcpScorer, synth

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import render_oracle as ro  # noqa: E402

pytestmark = pytest.mark.gpu

K = np.array([[154.0, 0, 80.3], [0, 153.2, 59.6], [0, 0, 1]], np.float32)     # a 160 x 120 camera
ROWS, COLS = 120, 160


def _cam_dict(z_near=0.1, z_max=1.0, K=K, rows=ROWS, cols=COLS):
    return dict(rows=rows, cols=cols, fx=float(K[0, 0]), fy=float(K[1, 1]), cx=float(K[0, 2]), cy=float(K[1, 2]),
                z_near=z_near, z_max=z_max)


def icosphere(level, radius, stretch=(1.0, 0.7, 0.5)):
    t = (1 + 5 ** 0.5) / 2
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6),
         (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.array(p, float) / np.linalg.norm(p) for p in v]
    for _ in range(level):
        cache, nf = {}, []

        def mid(a, b):
            k = (min(a, b), max(a, b))
            if k not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[k] = len(v) - 1
            return cache[k]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return (np.array(v) * radius * np.array(stretch)).astype(np.float32), np.array(f, np.int32)


def poses(rng, n, z=(0.35, 0.9)):
    return np.stack([synth.colmajor16(synth._se3(synth._random_rot(rng),
                                                 [rng.uniform(-0.12, 0.12), rng.uniform(-0.08, 0.08), rng.uniform(*z)]))
                     for _ in range(n)])


def test_point_splat_is_the_restatement_bit_for_bit():
    rng = np.random.default_rng(1)
    pts = synth.make_model(rng, 6000)[0].astype(np.float32)
    T = poses(rng, 6)
    sc = LcpScorer()
    cam = LcpScorer.camera(K, ROWS, COLS, 0.1, 1.0)
    got = sc.render_depth(pts, None, T, cam)
    for k in range(len(T)):
        want = ro.splat(pts, T[k], _cam_dict())
        assert np.array_equal(got[k].view(np.uint32), want.view(np.uint32)), k
        assert (got[k] > 0).sum() > 200
    # depths beyond z_max are dropped (renderScene.cpp:69), with a parent image underneath (UCTState.cpp:62-68)
    parent = np.zeros((ROWS, COLS), np.float32)
    parent[20:90, 30:120] = 0.6
    far = T.copy()
    far[:, 14] += 0.4        # some poses now straddle 1 m
    got = sc.render_depth(pts, None, far, cam, parent=parent)
    for k in range(len(T)):
        want = ro.splat(pts, far[k], _cam_dict(), parent=parent)
        assert np.array_equal(got[k].view(np.uint32), want.view(np.uint32)), k
        assert got[k].max() <= 1.0


@pytest.mark.parametrize("level", [1, 3])
def test_triangle_raster_is_the_restatement_bit_for_bit(level):
    rng = np.random.default_rng(2 + level)
    v, f = icosphere(level, 0.07)
    T = poses(rng, 5)
    T[4, 12:15] = [0.15, 0.0, 0.25]      # partly outside the image, large on screen
    sc = LcpScorer()
    cam = LcpScorer.camera(K, ROWS, COLS, 0.1, 1.0)
    got = sc.render_depth(v, f, T, cam)
    for k in range(len(T)):
        want = ro.raster(v, f, T[k], _cam_dict())
        assert np.array_equal(got[k].view(np.uint32), want.view(np.uint32)), (k, np.abs(got[k] - want).max())
    assert (got[0] > 0).sum() > 300
    # a closed surface: the rendered depth is the NEAR side -- the middle of the silhouette is in front of the centre
    c, drawn = T[0, 14], got[0][got[0] > 0]
    assert np.median(drawn) < c and drawn.min() >= c - 0.07 - 1e-3 and drawn.max() <= c + 0.07


def test_near_plane_clipping_and_view_filling_triangles():
    """a table quad that fills the view and runs past the camera (OpenGL clips, renderScene.cpp:45-72): its
    two triangles have vertices BEHIND the near plane and a pixel box of the whole image -- clipped against z_near
    (csrc/render.hip clip_edge) and filled by a workgroup each (render_big); bit-equal to oracle/render_oracle.py."""
    rng = np.random.default_rng(9)
    sc = LcpScorer()
    # a table: a 3 m x 3 m quad 0.25 m below the camera, tilted, reaching 1 m behind it; an object mesh on top
    quad = np.array([[-1.5, 0.25, -1.0], [1.5, 0.25, -1.0], [1.5, 0.25, 2.0], [-1.5, 0.25, 2.0]], np.float32)
    qf = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    ball, bf = icosphere(2, 0.06)
    ball = ball + np.array([0.02, 0.15, 0.55], np.float32)
    v = np.concatenate([quad, ball]).astype(np.float32)
    f = np.concatenate([qf, bf + 4]).astype(np.int32)
    T = np.stack([synth.colmajor16(synth._se3(synth._rot_axis_angle([1, 0.3, 0.1], a), [0.0, 0.02 * k, 0.0]))
                  for k, a in enumerate((0.0, 0.15, -0.2, 0.35))])
    for rows, cols, Kc in ((ROWS, COLS, K), (480, 640, np.array([[614.0, 0, 322.5], [0, 614.0, 239.7], [0, 0, 1]], np.float32))):
        for z_near in (0.1, 0.0):
            cam = LcpScorer.camera(Kc, rows, cols, z_near, 2.5)
            camd = _cam_dict(z_near=z_near, z_max=2.5, K=Kc, rows=rows, cols=cols)
            got = sc.render_depth(v, f, T, cam)
            for k in range(len(T) if rows == ROWS else 2):
                want = ro.raster(v, f, T[k], camd)
                assert np.array_equal(got[k].view(np.uint32), want.view(np.uint32)), (rows, z_near, k, np.abs(got[k] - want).max())
            # the table fills the lower part of the view; the ball is in front of it
            assert (got[0] > 0).mean() > 0.3 and got[0][got[0] > 0].min() < 0.55
    # every vertex order of a clipped triangle gives the same coverage
    tri = np.array([[0, 1, 2]], np.int32)
    cam = LcpScorer.camera(K, ROWS, COLS, 0.1, 2.5)
    a = sc.render_depth(quad, tri, T[:1], cam)[0]
    for perm in ([1, 2, 0], [2, 0, 1], [2, 1, 0]):
        b = sc.render_depth(quad, tri[:, perm], T[:1], cam)[0]
        assert np.array_equal(a > 0, b > 0) and np.allclose(a, b, atol=1e-6)


def test_edge_cases():
    sc = LcpScorer()
    cam = LcpScorer.camera(K, ROWS, COLS, 0.1, 1.0)
    v, f = icosphere(0, 0.05)
    T = poses(np.random.default_rng(3), 2)
    assert sc.render_depth(v, f, T[:0], cam).shape == (0, ROWS, COLS)
    behind = T.copy()
    behind[:, 14] = -0.5
    assert not sc.render_depth(v, f, behind, cam).any()                      # behind the camera: nothing
    assert not sc.render_depth(np.zeros((0, 3), np.float32), None, T, cam).any()   # no vertices: empty images
    bad = f.copy()
    bad[0, 0] = 9999                                                           # an index out of range is skipped
    a, b = sc.render_depth(v, bad, T, cam), sc.render_depth(v, f[1:], T, cam)
    assert np.array_equal(a, b)
    nanv = v.copy()
    nanv[3] = np.nan
    got = sc.render_depth(nanv, f, T, cam)
    want = ro.raster(nanv, f, T[0], _cam_dict())
    assert np.array_equal(got[0].view(np.uint32), want.view(np.uint32))


def test_leaf_states_rendered_and_costed_without_leaving_the_device():
    """64 leaf states at 640 x 480: render (point splat, the rule of tests/test_config5_real_frame_gpu.py) ->
    pgp_depth_cost_device, images never cross PCIe; the tallies equal pgp_depth_cost (host pointers) on images
    rendered by the numpy restatement of the same rule, and the restatement of computeCost."""
    import torch
    rng = np.random.default_rng(4)
    K640 = np.array([[614.0, 0, 322.5], [0, 614.0, 239.7], [0, 0, 1]], np.float32)
    rows, cols = 480, 640
    pts = synth.make_model(rng, 30000)[0].astype(np.float32)
    truth = synth._se3(synth._random_rot(rng), [0.03, -0.02, 0.62])
    n = 64
    T = np.stack([synth.colmajor16(truth @ synth._se3(synth._random_rot(rng, np.deg2rad(8)), 0.01 * rng.standard_normal(3)))
                  for _ in range(n)])
    camd = _cam_dict(K=K640, rows=rows, cols=cols)
    table = np.zeros((rows, cols), np.float32)
    table[300:, :] = 0.8                                        # the parent state's image: a far surface
    observed = ro.splat(pts, synth.colmajor16(truth), camd, parent=table)
    observed += (observed > 0) * rng.normal(0, 0.002, observed.shape).astype(np.float32)
    sc = LcpScorer()
    cam = LcpScorer.camera(K640, rows, cols, 0.1, 1.0)
    d_pts = torch.from_numpy(pts).cuda()
    d_T = torch.from_numpy(T).cuda()
    d_parent = torch.from_numpy(table).cuda()
    d_obs = torch.from_numpy(observed).cuda()
    d_img = sc.render_depth_device(d_pts, None, d_T, cam, d_parent=d_parent)
    d_counts, d_scores = sc.depth_cost_device(d_obs, d_img, 0.01)
    torch.cuda.synchronize()
    counts, scores = d_counts.cpu().numpy(), d_scores.cpu().numpy()
    host_imgs = np.stack([ro.splat(pts, T[k], camd, parent=table) for k in range(0, n, 8)])
    assert np.array_equal(d_img[::8].cpu().numpy().view(np.uint32), host_imgs.view(np.uint32))
    s_host, c_host = sc.depth_cost(observed, host_imgs, 0.01)
    assert np.array_equal(counts[::8], c_host) and np.array_equal(scores[::8], s_host)
    assert np.array_equal(c_host, ro.depth_cost(observed, host_imgs, 0.01))
    assert np.array_equal(scores, (counts[:, 0] + counts[:, 1] - counts[:, 2]).astype(np.float32))
    # one parent per image: image i laid over parent i
    d_par_n = d_parent[None].repeat(4, 1, 1).contiguous()
    d_par_n[1] = 0
    d4 = sc.render_depth_device(d_pts, None, d_T[:4].contiguous(), cam, d_parent=d_par_n)
    torch.cuda.synchronize()
    assert torch.equal(d4[0], d_img[0]) and not torch.equal(d4[1], d_img[1])
    assert np.array_equal(d4[1].cpu().numpy().view(np.uint32), ro.splat(pts, T[1], camd).view(np.uint32))
