#!/usr/bin/env python3
"""Fixture for the moving-least-squares step (Segmentation.cpp:239-246): tests/golden/mls.npz.

PCL is not vendored and not installed, so nothing here (or anywhere in this repository) runs PCL.  The
expected values come from an INDEPENDENT implementation of the published algorithm -- numpy float64:
neighbours by scipy cKDTree (then the strict float test of FLANN's radius search), plane by numpy.linalg.eigh
of the covariance, polynomial by numpy.linalg.lstsq on the sqrt-weighted design matrix -- that shares no
code and no numerical method with the C restatement (closed-form cubic roots, hand-written Cholesky) or the
HIP kernel.  The in-plane frame (u, v) follows Eigen's unitOrthogonal() as PCL uses it; the normal's SIGN
follows whatever the eigen-solver returns, so it is stored and compared up to sign.

Run from the repository root:  python tests/golden/make_mls_golden.py"""
import os

import numpy as np
from scipy.spatial import cKDTree

HERE = os.path.dirname(os.path.abspath(__file__))


def unit_orthogonal(n):
    if not (abs(n[0]) <= abs(n[2]) * 1e-12 and abs(n[1]) <= abs(n[2]) * 1e-12):
        inv = 1.0 / np.hypot(n[0], n[1])
        return np.array([-n[1] * inv, n[0] * inv, 0.0])
    inv = 1.0 / np.hypot(n[1], n[2])
    return np.array([0.0, -n[2] * inv, n[1] * inv])


def mls_numpy(xyz32, radius):
    P = xyz32.astype(np.float64)
    tree = cKDTree(P)
    r2 = np.float32(np.float64(radius) ** 2)
    out = []
    for i, p32 in enumerate(xyz32):
        cand = tree.query_ball_point(P[i], float(radius) * 1.01)
        d = xyz32[cand] - p32                                    # float32, as FLANN's L2_Simple
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        nb = np.sort(np.asarray(cand)[d2 < r2])
        if len(nb) < 3:
            continue
        Q = P[nb]
        c = Q.mean(axis=0)
        cov = (Q - c).T @ (Q - c)
        w, v = np.linalg.eigh(cov)
        nrm = v[:, 0]
        pt = P[i] - ((P[i] - c) @ nrm) * nrm
        tr = np.float32(np.trace(cov))
        curv = np.float32(0) if tr == 0 else np.float32(abs(np.float32(w[0] / np.float64(tr))))
        normal = nrm.copy()
        if len(nb) >= 6:
            va = unit_orthogonal(nrm)
            ua = np.cross(nrm, va)
            dm = Q - pt
            sq = np.einsum("ij,ij->i", dm, dm).astype(np.float32).astype(np.float64)
            wt = np.exp(-sq / (np.float64(radius) ** 2))
            u, vv, f = dm @ ua, dm @ va, dm @ nrm
            A = np.stack([np.ones_like(u), vv, vv * vv, u, u * vv, u * u], axis=1)
            sw = np.sqrt(wt)
            coef, *_ = np.linalg.lstsq(A * sw[:, None], f * sw, rcond=None)
            if np.isfinite(coef[0]) and np.linalg.matrix_rank(A * sw[:, None]) == 6:
                pt = pt + coef[0] * nrm
                normal = nrm - coef[3] * ua - coef[1] * va
        out.append((i, pt, normal, curv, len(nb)))
    idx = np.array([o[0] for o in out], np.int32)
    return (np.array([o[1] for o in out]), np.array([o[2] for o in out]), np.array([o[3] for o in out], np.float32), idx,
            np.array([o[4] for o in out], np.int32))


def make_cloud(seed=11):
    """a curved sheet (what a voxel-gridded segment looks like at 1 cm), a sparse tail whose points have
    3..5 neighbours (plane only), and isolated points (dropped)"""
    rng = np.random.default_rng(seed)
    g = np.stack(np.meshgrid(np.arange(-0.12, 0.12, 0.01), np.arange(-0.08, 0.08, 0.01), indexing="ij"), -1).reshape(-1, 2)
    g = g + rng.uniform(-0.004, 0.004, size=g.shape)
    z = 0.6 + 0.8 * g[:, 0] ** 2 - 0.5 * g[:, 0] * g[:, 1] + 0.3 * g[:, 1] ** 2 + rng.normal(0, 0.0008, len(g))
    sheet = np.column_stack([g, z])
    tail = np.column_stack([np.arange(0.16, 0.30, 0.013), np.zeros(11), np.full(11, 0.6)])[:11]
    tail = tail + rng.uniform(-0.002, 0.002, size=tail.shape)
    lone = np.array([[0.5, 0.5, 0.9], [-0.6, 0.1, 0.7], [0.5, 0.512, 0.9]])
    cloud = np.vstack([sheet, tail, lone]).astype(np.float32)
    return cloud[rng.permutation(len(cloud))]


if __name__ == "__main__":
    xyz = make_cloud()
    radius = np.float32(0.02)
    pts, nrm, curv, idx, cnt = mls_numpy(xyz, radius)
    np.savez_compressed(os.path.join(HERE, "mls.npz"), xyz=xyz, radius=radius, out_xyz=pts, out_nrm=nrm, out_curv=curv,
                        out_index=idx, n_neighbours=cnt)
    print(f"mls.npz: {len(xyz)} points in, {len(idx)} out, neighbours {cnt.min()}..{cnt.max()}, "
          f"{int((cnt < 6).sum())} plane-only")
