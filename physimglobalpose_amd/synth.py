"""Seeded synthetic workloads for the pose-hypothesis scoring path.

Nothing here touches the reference or the oracle: it only manufactures inputs of the shape the
path consumes at its boundary (base.cc:1885-1901): a centred scene cloud P with unit normals and
per-point weights, a centred validation model Q_val with unit normals, and a list of centred
4x4 float transforms (the memory image of the reference's `allTransforms`, column-major).

The generator is numpy-only so that tests, bench.py and tests/golden/make_golden.py share it.
"""
from __future__ import annotations

import dataclasses
import numpy as np

SEED_BASE = 0xC0FFEE  # + config id
DELTA = 0.005         # S4/super4pcs_test.cc:20
GATE_DEG = 30.0       # base.cc:1758


@dataclasses.dataclass
class Workload:
    """One object's scoring problem, already in the reference's centred frames."""
    P_xyz: np.ndarray      # (nP,3) f32 scene, centred on centroid_P
    P_nrm: np.ndarray      # (nP,3) f32 unit normals
    P_w: np.ndarray        # (nP,)  f32 weights (orig_probabilities_)
    Q_xyz: np.ndarray      # (nQ,3) f32 validation model, centred on centroid_Q
    Q_nrm: np.ndarray      # (nQ,3) f32 unit normals
    Qs_xyz: np.ndarray     # (nQs,3) f32 sparse search model (congruent-set side), centred
    T: np.ndarray          # (nH,16) f32 column-major centred transforms
    T_gt: np.ndarray       # (16,) f32 the ground-truth centred transform
    centroid_P: np.ndarray
    centroid_Q: np.ndarray
    delta: float = DELTA
    gate_deg: float = GATE_DEG
    Qs_nrm: np.ndarray = None       # (nQs,3) f32 normals of the search model
    T_gt_world: np.ndarray = None   # (4,4) f64 ground-truth pose, model frame -> camera frame

    @property
    def n_h(self) -> int:
        return int(self.T.shape[0])


def _unit(v):
    n = np.linalg.norm(v, axis=-1, keepdims=True)
    return v / np.where(n == 0, 1.0, n)


def _rot_axis_angle(axis, ang):
    axis = _unit(np.asarray(axis, dtype=np.float64))
    x, y, z = axis
    c, s = np.cos(ang), np.sin(ang)
    C = 1 - c
    return np.array([[c + x * x * C, x * y * C - z * s, x * z * C + y * s],
                     [y * x * C + z * s, c + y * y * C, y * z * C - x * s],
                     [z * x * C - y * s, z * y * C + x * s, c + z * z * C]])


def _random_rot(rng, max_angle=None):
    """Uniform SO(3) if max_angle is None, else random axis with angle U(0,max_angle)."""
    if max_angle is None:
        q = rng.standard_normal(4)
        q /= np.linalg.norm(q)
        w, x, y, z = q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    return _rot_axis_angle(rng.standard_normal(3), rng.uniform(0, max_angle))


def _se3(R, t):
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    return T


def _sample_box(rng, n, size, center=(0, 0, 0)):
    """Area-weighted points + outward face normals on an axis-aligned box surface."""
    sx, sy, sz = size
    areas = np.array([sy * sz, sy * sz, sx * sz, sx * sz, sx * sy, sx * sy])
    face = rng.choice(6, size=n, p=areas / areas.sum())
    u = rng.uniform(-0.5, 0.5, size=(n, 2))
    pts = np.zeros((n, 3))
    nrm = np.zeros((n, 3))
    for f in range(6):
        m = face == f
        ax = f // 2
        sign = 1.0 if f % 2 == 0 else -1.0
        others = [a for a in range(3) if a != ax]
        pts[m, ax] = sign * 0.5 * size[ax]
        pts[m, others[0]] = u[m, 0] * size[others[0]]
        pts[m, others[1]] = u[m, 1] * size[others[1]]
        nrm[m, ax] = sign
    return pts + np.asarray(center), nrm


def _sample_cylinder_side(rng, n, r, h, center):
    th = rng.uniform(0, 2 * np.pi, n)
    z = rng.uniform(-0.5 * h, 0.5 * h, n)
    nrm = np.stack([np.cos(th), np.sin(th), np.zeros(n)], axis=1)
    pts = nrm * r + np.stack([np.zeros(n), np.zeros(n), z], axis=1) + np.asarray(center)
    return pts, nrm


def _sample_sphere(rng, n, r, center):
    d = _unit(rng.standard_normal((n, 3)))
    return d * r + np.asarray(center), d


def make_model(rng, n):
    """A 0.20 x 0.12 x 0.08 m box with a 3 cm-radius, 6 cm-tall knob on its top face."""
    box = (0.20, 0.12, 0.08)
    a_box = 2 * (box[0] * box[1] + box[1] * box[2] + box[0] * box[2])
    r, h = 0.03, 0.06
    a_cyl = 2 * np.pi * r * h + np.pi * r * r
    n_cyl = int(round(n * a_cyl / (a_box + a_cyl)))
    n_cap = int(round(n_cyl * (np.pi * r * r) / a_cyl))
    n_side = n_cyl - n_cap
    n_box = n - n_cyl
    pb, nb = _sample_box(rng, n_box, box)
    cz = 0.5 * box[2] + 0.5 * h
    ps, ns = _sample_cylinder_side(rng, n_side, r, h, (0.04, 0.01, cz))
    rad = r * np.sqrt(rng.uniform(0, 1, n_cap))
    th = rng.uniform(0, 2 * np.pi, n_cap)
    pc = np.stack([0.04 + rad * np.cos(th), 0.01 + rad * np.sin(th), np.full(n_cap, cz + 0.5 * h)], axis=1)
    nc = np.tile(np.array([0.0, 0.0, 1.0]), (n_cap, 1))
    pts = np.concatenate([pb, ps, pc])
    nrm = np.concatenate([nb, ns, nc])
    perm = rng.permutation(n)
    return pts[perm], nrm[perm]


def _farthest_subset(pts, k, start=0):
    n = pts.shape[0]
    k = min(k, n)
    sel = np.empty(k, dtype=np.int64)
    sel[0] = start
    d = np.linalg.norm(pts - pts[start], axis=1)
    for i in range(1, k):
        j = int(np.argmax(d))
        sel[i] = j
        d = np.minimum(d, np.linalg.norm(pts - pts[j], axis=1))
    return sel


def _perturb_normals(rng, nrm, sigma_deg):
    ang = np.deg2rad(sigma_deg)
    noisy = nrm + ang * rng.standard_normal(nrm.shape)
    return _unit(noisy)


def _center_f32(P, Qs, Qv):
    """base.cc:242-268 in float32: sequential float accumulation, one division, subtraction."""
    def centroid(X):
        c = np.zeros(3, dtype=np.float32)
        for k in range(3):
            acc = np.float32(0)
            col = X[:, k]
            # sequential float32 accumulation (np.cumsum on f32 is sequential)
            acc = np.cumsum(col, dtype=np.float32)[-1] if len(col) else np.float32(0)
            c[k] = acc / np.float32(len(col))
        return c
    cP = centroid(P)
    cQ = centroid(Qs)
    return (P - cP).astype(np.float32), (Qs - cQ).astype(np.float32), (Qv - cQ).astype(np.float32), cP, cQ


def make_workload(n_scene=50000, n_model=5000, n_hyp=4096, config_id=2, n_search=500,
                  seed=None) -> Workload:
    """The C2-shaped workload of SURVEY 8d at arbitrary sizes (all seeded)."""
    rng = np.random.default_rng(SEED_BASE + config_id if seed is None else seed)
    Qw, Qn = make_model(rng, n_model)

    # ground-truth pose: model -> camera frame, z in [0.3, 1.2]
    R_gt = _rot_axis_angle([0.3, -0.8, 0.5], 1.1)
    t_gt = np.array([0.05, -0.03, 0.80])
    obj = Qw @ R_gt.T + t_gt
    obj_n = Qn @ R_gt.T
    facing = np.einsum("ij,ij->i", obj_n, -_unit(obj)) > 0.05   # camera at the origin
    keep = np.flatnonzero(facing)
    max_obj = max(1, n_scene // 10)
    if keep.size > max_obj:
        keep = keep[:max_obj]
    obj = obj[keep] + 0.001 * rng.standard_normal((keep.size, 3))
    obj_n = _perturb_normals(rng, obj_n[keep], 3.0)

    rest = n_scene - keep.size
    n_table = rest // 2
    n_dis = (rest * 3) // 10
    n_out = rest - n_table - n_dis
    parts_p, parts_n = [obj], [obj_n]
    # table patch: plane through the lowest object point, normal facing the camera-ish
    up = _unit(R_gt @ np.array([0.0, 0.0, 1.0]))
    e1 = _unit(np.cross(up, [1.0, 0.0, 0.0]))
    e2 = np.cross(up, e1)
    base = t_gt - up * 0.045
    uv = rng.uniform(-0.35, 0.35, size=(n_table, 2))
    tab = base + uv[:, :1] * e1 + uv[:, 1:] * e2 + 0.001 * rng.standard_normal((n_table, 3))
    parts_p.append(tab)
    parts_n.append(_perturb_normals(rng, np.tile(up, (n_table, 1)), 3.0))
    # four distractor solids standing on the table
    per = [n_dis // 4] * 4
    per[-1] += n_dis - sum(per)
    offs = [(0.22, 0.10), (-0.20, 0.15), (0.15, -0.22), (-0.18, -0.17)]
    for k, (a, b) in enumerate(offs):
        c = base + a * e1 + b * e2 + up * 0.05
        if k % 2 == 0:
            p, nn = _sample_sphere(rng, per[k], 0.05, c)
        else:
            p, nn = _sample_box(rng, per[k], (0.10, 0.08, 0.10))
            Rk = _random_rot(rng)
            p, nn = p @ Rk.T + c, nn @ Rk.T
        parts_p.append(p + 0.001 * rng.standard_normal(p.shape))
        parts_n.append(_perturb_normals(rng, nn, 3.0))
    lo = np.array([-0.45, -0.45, 0.30])
    hi = np.array([0.55, 0.45, 1.20])
    parts_p.append(rng.uniform(lo, hi, size=(n_out, 3)))
    parts_n.append(_unit(rng.standard_normal((n_out, 3))))
    Pw = np.concatenate(parts_p)
    Pn = np.concatenate(parts_n)
    w = rng.uniform(0.0, 0.3, size=Pw.shape[0])
    w[:keep.size] = 1.0
    perm = rng.permutation(Pw.shape[0])
    Pw, Pn, w = Pw[perm], Pn[perm], w[perm]

    sel = _farthest_subset(Qw, n_search)
    P32 = Pw.astype(np.float32)
    Qv32 = Qw.astype(np.float32)
    Qs32 = Qv32[sel]
    Pc, Qsc, Qvc, cP, cQ = _center_f32(P32, Qs32, Qv32)

    # hypotheses, world frame (double), then centred: T_c = tr(-cP) * T * tr(+cQ)
    T_gt = _se3(R_gt, t_gt)
    n_small = n_hyp // 4
    n_med = n_hyp // 2
    n_rand = n_hyp - n_small - n_med
    Ts = []
    for _ in range(n_small):
        d = _se3(_random_rot(rng, np.deg2rad(5.0)), 0.003 * rng.standard_normal(3))
        Ts.append(_se3(np.eye(3), t_gt) @ d @ _se3(R_gt, np.zeros(3)))
    for _ in range(n_med):
        d = _se3(_random_rot(rng, np.deg2rad(30.0)), 0.02 * rng.standard_normal(3))
        Ts.append(_se3(np.eye(3), t_gt) @ d @ _se3(R_gt, np.zeros(3)))
    for _ in range(n_rand):
        Ts.append(_se3(_random_rot(rng), rng.uniform(lo, hi)))
    Ts = np.stack(Ts) if Ts else np.zeros((0, 4, 4))
    order = rng.permutation(n_hyp)
    Ts = Ts[order]
    A = _se3(np.eye(3), -cP.astype(np.float64))
    B = _se3(np.eye(3), cQ.astype(np.float64))

    def centred(T):
        Tc = (A @ T @ B).astype(np.float32)
        return np.ascontiguousarray(Tc.T).reshape(16)  # column-major image

    Tc = np.stack([centred(T) for T in Ts]) if n_hyp else np.zeros((0, 16), np.float32)
    return Workload(
        P_xyz=np.ascontiguousarray(Pc), P_nrm=np.ascontiguousarray(Pn.astype(np.float32)),
        P_w=np.ascontiguousarray(w.astype(np.float32)),
        Q_xyz=np.ascontiguousarray(Qvc), Q_nrm=np.ascontiguousarray(Qn.astype(np.float32)),
        Qs_xyz=np.ascontiguousarray(Qsc), T=np.ascontiguousarray(Tc.astype(np.float32)),
        T_gt=centred(T_gt), centroid_P=cP, centroid_Q=cQ,
        Qs_nrm=np.ascontiguousarray(Qn[sel].astype(np.float32)), T_gt_world=T_gt)


def colmajor16(T4x4) -> np.ndarray:
    """4x4 (row-major numpy) -> the 16-float column-major image the C-ABI takes."""
    return np.ascontiguousarray(np.asarray(T4x4, dtype=np.float32).T).reshape(16)
