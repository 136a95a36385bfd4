"""ctypes bindings to the CHECKERS: oracle/liboracle.so (our C restatement) and, when built,
oracle/_ref/libpgp_ref.so (harness over the reference's own kd-tree + Eigen).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libpgp_ref.so")

_f = C.POINTER(C.c_float)
_i = C.POINTER(C.c_int)
_d = C.POINTER(C.c_double)


def _fp(a):
    return None if a is None else a.ctypes.data_as(_f)


def _ip(a):
    return None if a is None else a.ctypes.data_as(_i)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def build_oracle():
    """Compile the C restatement if missing or stale (gcc, < 1 s)."""
    src = os.path.join(ORACLE_DIR, "pgp_oracle.c")
    if (not os.path.exists(ORACLE_SO)) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)
    return ORACLE_SO


_oracle = None


def oracle_lib():
    global _oracle
    if _oracle is None:
        L = C.CDLL(build_oracle())
        L.orc_kd_build.restype = C.c_void_p
        L.orc_kd_build.argtypes = [_f, C.c_int]
        L.orc_kd_free.argtypes = [C.c_void_p]
        L.orc_kd_num_nodes.argtypes = [C.c_void_p]
        L.orc_kd_query.argtypes = [C.c_void_p, _f, C.c_float]
        L.orc_brute_query.argtypes = [_f, C.c_int, _f, C.c_float]
        L.orc_transform_point.argtypes = [_f, _f, _f]
        L.orc_rotate_normal.argtypes = [_f, _f, _f]
        L.orc_sqdist.restype = C.c_float
        L.orc_sqdist.argtypes = [_f, _f]
        L.orc_dot.restype = C.c_float
        L.orc_dot.argtypes = [_f, _f]
        L.orc_normal_gate.argtypes = [C.c_float, C.c_float]
        L.orc_verify.restype = C.c_float
        L.orc_verify.argtypes = [C.c_void_p, _f, C.c_int, _f, C.c_int, _f, C.c_float, C.c_float,
                                 C.c_int, _i, _i]
        L.orc_weighted_verify.restype = C.c_float
        L.orc_weighted_verify.argtypes = [C.c_void_p, _f, _f, _f, C.c_int, _f, _f, C.c_int, _f,
                                          C.c_float, C.c_float, _i, _i]
        L.orc_score_batch.argtypes = [C.c_void_p, _f, _f, _f, C.c_int, _f, _f, C.c_int, _f, C.c_int,
                                      C.c_float, C.c_int, C.c_float, C.c_int, C.c_int, _f, _i, _i, _i]
        L.orc_center.argtypes = [_f, C.c_int, _f, C.c_int, _f, C.c_int, _f, _f]
        L.orc_rigid_from_pair.argtypes = [_f, _f, _f, _f, _f, _d, _f]
        L.orc_cs_create.restype = C.c_void_p
        L.orc_cs_create.argtypes = [_f, C.c_int]
        L.orc_cs_free.argtypes = [C.c_void_p]
        L.orc_cs_extract_pairs.argtypes = [C.c_void_p, C.c_float, C.c_float, _i, C.c_int]
        L.orc_cs_find_congruent.argtypes = [C.c_void_p, _f, C.c_float, C.c_float, C.c_float, _i, C.c_int,
                                            _i, C.c_int, _i, C.c_int]
        L.orc_icp.argtypes = [_f, C.c_int, _f, C.c_int, _f, C.c_int, C.c_float, C.c_float, C.c_float, _f]
        L.orc_pose_error.argtypes = [_f, _f, _f, _f, _f]
        L.orc_greedy_cluster.argtypes = [_f, _f, C.c_int, C.c_float, C.c_float, _f, C.c_float, C.c_float, _i, _i]
        L.orc_backproject.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_ubyte), C.c_int, C.c_int, _f, C.c_double,
                                      C.c_double, _f]
        L.orc_voxel_grid.argtypes = [_f, C.c_int, C.c_float, _f, C.c_int]
        L.orc_pose_hausdorff.argtypes = [_f, C.c_int, _f, _f, _f, _f]
        L.orc_mls.argtypes = [_f, C.c_int, C.c_float, _f, _f, _f, _i, C.c_int]
        _oracle = L
    return _oracle


def oracle_voxel_grid(xyz, leaf=0.01):
    xyz = _f32(xyz).reshape(-1, 3)
    out = np.zeros((max(len(xyz), 1), 3), np.float32)
    n = oracle_lib().orc_voxel_grid(_fp(xyz), len(xyz), C.c_float(leaf), _fp(out), len(xyz))
    return out[:n].copy()


def oracle_mls(xyz, radius=0.02):
    """C restatement of pcl::MovingLeastSquares (order 2, normals): (xyz, normals, curvature, input index)."""
    xyz = _f32(xyz).reshape(-1, 3)
    n = max(len(xyz), 1)
    ox, on, oc, oi = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), np.zeros(n, np.float32), np.zeros(n, np.int32)
    m = oracle_lib().orc_mls(_fp(xyz), len(xyz), C.c_float(radius), _fp(ox), _fp(on), _fp(oc), oi.ctypes.data_as(_i), n)
    return ox[:m].copy(), on[:m].copy(), oc[:m].copy(), oi[:m].copy()


def _hausdorff(fn, hull, T, pairs):
    hull, T = _f32(hull).reshape(-1, 3), _f32(T).reshape(-1, 16)
    dmax, dsum = np.zeros(len(pairs), np.float32), np.zeros(len(pairs), np.float32)
    a, b = C.c_float(0), C.c_float(0)
    for k, (i, j) in enumerate(pairs):
        fn(_fp(hull), len(hull), _fp(np.ascontiguousarray(T[i])), _fp(np.ascontiguousarray(T[j])), C.byref(a), C.byref(b))
        dmax[k], dsum[k] = a.value, b.value
    return dmax, dsum


def oracle_pose_hausdorff(hull, T, pairs):
    return _hausdorff(oracle_lib().orc_pose_hausdorff, hull, T, pairs)


def ref_pose_hausdorff(hull, T, pairs):
    L = ref_lib()
    L.ref_pose_hausdorff.argtypes = [_f, C.c_int, _f, _f, _f, _f]
    return _hausdorff(L.ref_pose_hausdorff, hull, T, pairs)


class Oracle:
    """The C restatement bound to one (P, Q_val) pair.  use_kd=False -> brute-force NN."""

    def __init__(self, P_xyz, P_nrm, P_w, Q_xyz, Q_nrm, use_kd=True):
        self.L = oracle_lib()
        self.P = _f32(P_xyz)
        self.Pn = _f32(P_nrm)
        self.Pw = _f32(P_w)
        self.Q = _f32(Q_xyz)
        self.Qn = _f32(Q_nrm)
        self.nP, self.nQ = self.P.shape[0], self.Q.shape[0]
        self.kd = self.L.orc_kd_build(_fp(self.P), self.nP) if use_kd else None

    def __del__(self):
        if getattr(self, "kd", None):
            self.L.orc_kd_free(self.kd)
            self.kd = None

    def kd_query(self, q, sqdist):
        q = _f32(q)
        return self.L.orc_kd_query(self.kd, _fp(q), C.c_float(sqdist))

    def brute_query(self, q, sqdist):
        q = _f32(q)
        return self.L.orc_brute_query(_fp(self.P), self.nP, _fp(q), C.c_float(sqdist))

    def verify(self, T16, delta, best_lcp=0.0, early_out=False):
        T16 = _f32(T16)
        good = C.c_int(0)
        hits = np.full(self.nQ, -2, dtype=np.int32)
        s = self.L.orc_verify(self.kd, _fp(self.P), self.nP, _fp(self.Q), self.nQ, _fp(T16),
                              C.c_float(delta), C.c_float(best_lcp), int(early_out),
                              C.byref(good), _ip(hits))
        return float(np.float32(s)), good.value, hits

    def weighted_verify(self, T16, delta, gate_deg=30.0):
        T16 = _f32(T16)
        reg = np.zeros(max(self.nQ, 1), dtype=np.int32)
        n = C.c_int(0)
        s = self.L.orc_weighted_verify(self.kd, _fp(self.P), _fp(self.Pn), _fp(self.Pw), self.nP,
                                       _fp(self.Q), _fp(self.Qn), self.nQ, _fp(T16),
                                       C.c_float(delta), C.c_float(gate_deg), _ip(reg), C.byref(n))
        return float(np.float32(s)), reg[: n.value].copy()

    def score_batch(self, T, delta, mode=0, gate_deg=30.0, early_out=False, threads=1):
        T = _f32(T).reshape(-1, 16)
        n_h = T.shape[0]
        scores = np.zeros(n_h, dtype=np.float32)
        best = C.c_int(-1)
        sel = np.zeros(max(n_h, 1), dtype=np.int32)
        nsel = C.c_int(0)
        self.L.orc_score_batch(self.kd, _fp(self.P), _fp(self.Pn), _fp(self.Pw), self.nP,
                               _fp(self.Q), _fp(self.Qn), self.nQ, _fp(T), n_h, C.c_float(delta),
                               int(mode), C.c_float(gate_deg), int(early_out), int(threads),
                               _fp(scores), C.byref(best), _ip(sel), C.byref(nsel))
        return scores, best.value, sel[: nsel.value].copy()


def oracle_rigid_from_pairs(P_xyz, Qs_xyz, base_ids, quad_ids, cP, cQ):
    """orc_rigid_from_pair over a batch: returns (T, pose, status, rms) like the C ABI."""
    L = oracle_lib()
    P, Qs = _f32(P_xyz), _f32(Qs_xyz)
    cP, cQ = _f32(cP), _f32(cQ)
    n = len(base_ids)
    T = np.full((n, 16), np.nan, np.float32)
    pose = np.full((n, 16), np.nan, np.float64)
    status = np.zeros(n, np.int32)
    rms = np.zeros(n, np.float32)
    for i in range(n):
        p = np.ascontiguousarray(P[np.asarray(base_ids[i])])
        q = np.ascontiguousarray(Qs[np.asarray(quad_ids[i])])
        t16, p16, r = np.zeros(16, np.float32), np.zeros(16, np.float64), C.c_float(0)
        status[i] = L.orc_rigid_from_pair(_fp(p), _fp(q), _fp(cP), _fp(cQ), _fp(t16),
                                          p16.ctypes.data_as(_d), C.byref(r))
        rms[i] = r.value
        if status[i] == 1:
            T[i], pose[i] = t16, p16
    return T, pose, status, rms


class CongruentChecker:
    """Congruent-set extraction over one search model: kind='oracle' (C restatement) or 'ref'
    (the reference's own PairCreationFunctor / IntersectionFunctor / IndexedNormalSet)."""

    def __init__(self, Qs_xyz, kind="oracle", delta=0.005):
        self.kind = kind
        self.Qs = _f32(Qs_xyz)
        if kind == "oracle":
            self.L = oracle_lib()
            self.h = self.L.orc_cs_create(_fp(self.Qs), len(self.Qs))
        else:
            self.L = ref_lib()
            self.h = self.L.ref_cs_create(_fp(self.Qs), len(self.Qs), C.c_double(delta))

    def __del__(self):
        if getattr(self, "h", None):
            (self.L.orc_cs_free if self.kind == "oracle" else self.L.ref_cs_destroy)(self.h)
            self.h = None

    def extract_pairs(self, pair_distance, eps, base=None, cap=1 << 22):
        out = np.zeros((cap, 2), np.int32)
        if self.kind == "oracle":
            n = self.L.orc_cs_extract_pairs(self.h, C.c_float(pair_distance), C.c_float(eps), _ip(out), cap)
        else:
            b = _f32(base if base is not None else np.zeros((4, 3))).reshape(12)
            n = self.L.ref_cs_extract_pairs(self.h, _fp(b), 0, 1, C.c_float(pair_distance), C.c_float(eps),
                                            _ip(out), cap)
        assert n <= cap
        return out[:n].copy()

    def find_congruent(self, base, inv1, inv2, threshold, P_pairs, Q_pairs, cap=1 << 22):
        b = _f32(base).reshape(12)
        Pp = np.ascontiguousarray(P_pairs, np.int32).reshape(-1, 2)
        Qp = np.ascontiguousarray(Q_pairs, np.int32).reshape(-1, 2)
        out = np.zeros((cap, 4), np.int32)
        fn = self.L.orc_cs_find_congruent if self.kind == "oracle" else self.L.ref_cs_find_congruent
        n = fn(self.h, _fp(b), C.c_float(inv1), C.c_float(inv2), C.c_float(threshold), _ip(Pp), len(Pp),
               _ip(Qp), len(Qp), _ip(out), cap)
        assert n <= cap
        return out[:n].copy()


def oracle_icp(src, tgt, T, trim=1.0, max_iterations=100, max_corr_dist=0.0, energy_ratio=1.0):
    """orc_icp over a batch of guesses: returns (T_refined, energy, iters)."""
    L = oracle_lib()
    src, tgt = _f32(src), _f32(tgt)
    T = np.array(_f32(T).reshape(-1, 16), copy=True)
    energy = np.zeros(len(T), np.float32)
    iters = np.zeros(len(T), np.int32)
    for h in range(len(T)):
        e = C.c_float(0)
        iters[h] = L.orc_icp(_fp(src), len(src), _fp(tgt), len(tgt), _fp(T[h]), int(max_iterations),
                             C.c_float(trim), C.c_float(max_corr_dist), C.c_float(energy_ratio), C.byref(e))
        energy[h] = e.value
    return T, energy, iters


def _pose_error(fn, test, gt, sym):
    test, gt = _f32(test).reshape(-1, 16), _f32(gt).reshape(-1, 16)
    sym = _f32(sym)
    rot, trans = np.zeros(len(test), np.float32), np.zeros(len(test), np.float32)
    r, t = C.c_float(0), C.c_float(0)
    for i in range(len(test)):
        fn(_fp(test[i]), _fp(gt[i]), _fp(sym), C.byref(r), C.byref(t))
        rot[i], trans[i] = r.value, t.value
    return rot, trans


def oracle_pose_error(test, gt, sym=(0, 0, 0)):
    """orc_pose_error over n pairs of 16-float col-major transforms -> (rot_err_deg, trans_err)."""
    return _pose_error(oracle_lib().orc_pose_error, test, gt, sym)


def ref_pose_error(test, gt, sym=(0, 0, 0)):
    return _pose_error(ref_lib().ref_pose_error, test, gt, sym)


def oracle_greedy_cluster(T, scores, best_score, sym=(0, 0, 0), accept_fraction=0.5, rot_thresh=10.0,
                          trans_thresh=0.02):
    """orc_greedy_cluster -> (representative ids in output order, assignment per hypothesis)."""
    T, scores, sym = _f32(T).reshape(-1, 16), _f32(scores), _f32(sym)
    n = len(T)
    rep = np.zeros(max(n, 1), np.int32)
    assign = np.zeros(max(n, 1), np.int32)
    k = oracle_lib().orc_greedy_cluster(_fp(T), _fp(scores), n, C.c_float(best_score), C.c_float(accept_fraction),
                                        _fp(sym), C.c_float(rot_thresh), C.c_float(trans_thresh), _ip(rep),
                                        _ip(assign))
    return rep[:k].copy(), assign[:n].copy()


def ref_greedy_cluster(T, scores, best_score, sym=(0, 0, 0)):
    """The harness restatement of greedyClustering with the reference's std::sort (0.5 / 10 / 0.02)."""
    T, scores, sym = _f32(T).reshape(-1, 16), _f32(scores), _f32(sym)
    rep = np.zeros(max(len(T), 1), np.int32)
    k = ref_lib().ref_greedy_cluster(_fp(T), _fp(scores), len(T), C.c_float(best_score), _fp(sym), _ip(rep))
    return rep[:k].copy()


def _bp_args(image, mask, K):
    image = np.ascontiguousarray(image)
    assert image.dtype in (np.uint16, np.float32) and image.ndim == 2
    m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
    return image, m, _f32(K).reshape(9)


def oracle_backproject(image, mask, K, z_min=0.1, z_max=2.0):
    """orc_backproject: image (rows, cols) uint16 raw samples or float32 metres -> (n, 3) cloud."""
    image, m, K9 = _bp_args(image, mask, K)
    rows, cols = image.shape
    out = np.zeros((max(rows * cols, 1), 3), np.float32)
    n = oracle_lib().orc_backproject(image.ctypes.data_as(C.c_void_p), int(image.dtype == np.uint16),
                                     None if m is None else m.ctypes.data_as(C.POINTER(C.c_ubyte)), rows, cols,
                                     _fp(K9), C.c_double(z_min), C.c_double(z_max), _fp(out))
    return out[:n].copy()


def ref_backproject(raw16, mask, K):
    """The Eigen-typed harness restatement: decode (utilities.cpp:47-61) + back-project (:190-206)."""
    raw16, m, K9 = _bp_args(raw16, mask, K)
    rows, cols = raw16.shape
    L = ref_lib()
    depth = np.zeros(rows * cols, np.float32)
    L.ref_decode_depth(raw16.ctypes.data_as(C.POINTER(C.c_ushort)), rows * cols, _fp(depth))
    out = np.zeros((max(rows * cols, 1), 3), np.float32)
    n = L.ref_backproject(_fp(depth), None if m is None else m.ctypes.data_as(C.POINTER(C.c_ubyte)), rows, cols,
                          _fp(K9), _fp(out))
    return depth.reshape(rows, cols), out[:n].copy()


def have_ref():
    return os.path.exists(REF_SO)


_ref = None


def ref_lib():
    global _ref
    if _ref is None:
        L = C.CDLL(REF_SO)
        L.ref_create.restype = C.c_void_p
        L.ref_create.argtypes = [_f, _f, _f, C.c_int, _f, _f, C.c_int]
        L.ref_destroy.argtypes = [C.c_void_p]
        L.ref_get_normals.argtypes = [C.c_void_p, C.c_int, _f]
        L.ref_center.argtypes = [_f, C.c_int, _f, C.c_int, _f, C.c_int, _f, _f]
        L.ref_kd_query.argtypes = [C.c_void_p, _f, C.c_float]
        L.ref_verify.restype = C.c_float
        L.ref_verify.argtypes = [C.c_void_p, _f, C.c_float, C.c_float, C.c_int, _i, _i]
        L.ref_weighted_verify.restype = C.c_float
        L.ref_weighted_verify.argtypes = [C.c_void_p, _f, C.c_float, _i, _i]
        L.ref_transform_point.argtypes = [_f, _f, _f]
        L.ref_rotate_normal.argtypes = [_f, _f, _f]
        L.ref_sqdist.restype = C.c_float
        L.ref_sqdist.argtypes = [_f, _f]
        L.ref_dot.restype = C.c_float
        L.ref_dot.argtypes = [_f, _f]
        L.ref_rigid_from_pair.argtypes = [_f, _f, _f, _f, _f, _d, _f]
        L.ref_weights_from_image.argtypes = [_f, C.c_int, _f, _f, C.POINTER(C.c_ushort), C.c_int, C.c_int, _f]
        L.ref_cs_create.restype = C.c_void_p
        L.ref_cs_create.argtypes = [_f, C.c_int, C.c_double]
        L.ref_cs_destroy.argtypes = [C.c_void_p]
        L.ref_cs_extract_pairs.argtypes = [C.c_void_p, _f, C.c_int, C.c_int, C.c_float, C.c_float, _i, C.c_int]
        L.ref_cs_find_congruent.argtypes = [C.c_void_p, _f, C.c_float, C.c_float, C.c_float, _i, C.c_int,
                                            _i, C.c_int, _i, C.c_int]
        L.ref_pose_error.argtypes = [_f, _f, _f, _f, _f]
        L.ref_greedy_cluster.argtypes = [_f, _f, C.c_int, C.c_float, _f, _i]
        L.ref_decode_depth.argtypes = [C.POINTER(C.c_ushort), C.c_int, _f]
        L.ref_backproject.argtypes = [_f, C.POINTER(C.c_ubyte), C.c_int, C.c_int, _f, _f]
        L.ref_stocs_create.restype = C.c_void_p
        L.ref_stocs_create.argtypes = [_f, _f, _f, C.c_int, _i, C.c_int]
        L.ref_stocs_destroy.argtypes = [C.c_void_p]
        L.ref_stocs_get_normals.argtypes = [C.c_void_p, _f]
        L.ref_stocs_ppf.argtypes = [C.c_void_p, C.c_int, C.c_int, _i]
        L.ref_stocs_stage.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, _f, _f]
        L.ref_try_quadrilateral.argtypes = [C.c_void_p, _i, _f, _f]
        _ref = L
    return _ref


class RefStocs:
    """Base selection through the Eigen-typed harness (computePPF, the three weighting loops of
    SelectQuadrilateralStoCS, TryQuadrilateral).  Build container only."""

    def __init__(self, P_xyz, P_nrm, prob, keys):
        self.L = ref_lib()
        self.P, self.N, self.prob = _f32(P_xyz), _f32(P_nrm), _f32(prob)
        self.keys = np.ascontiguousarray(keys, np.int32).reshape(-1, 4)
        self.n = len(self.P)
        self.h = self.L.ref_stocs_create(_fp(self.P), _fp(self.N), _fp(self.prob), self.n, _ip(self.keys), len(self.keys))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.ref_stocs_destroy(self.h)
            self.h = None

    def normals(self):
        out = np.zeros((self.n, 3), np.float32)
        self.L.ref_stocs_get_normals(self.h, _fp(out))
        return out

    def ppf(self, i, j):
        f = np.zeros(4, np.int32)
        self.L.ref_stocs_ppf(self.h, int(i), int(j), _ip(f))
        return f

    def stage(self, stage, cur, b1, b2=-1, b3=-1):
        cur = np.array(_f32(cur), copy=True)
        s = C.c_float(0)
        present = self.L.ref_stocs_stage(self.h, int(stage), int(b1), int(b2), int(b3), _fp(cur), C.byref(s))
        return cur, float(np.float32(s.value)), bool(present)

    def try_quadrilateral(self, ids):
        ids = np.array(ids, np.int32)
        a, b = C.c_float(0), C.c_float(0)
        ok = self.L.ref_try_quadrilateral(self.h, _ip(ids), C.byref(a), C.byref(b))
        return ids, np.float32(a.value), np.float32(b.value), bool(ok)


class Ref:
    """The reference's own kd-tree (+ Eigen-order loop bodies) bound to one (P, Q_val) pair.
    Only usable in the build container (needs oracle/_ref built from /root/reference)."""

    def __init__(self, P_xyz, P_nrm, P_w, Q_xyz, Q_nrm):
        self.L = ref_lib()
        self.P, self.Q = _f32(P_xyz), _f32(Q_xyz)
        self.nP, self.nQ = self.P.shape[0], self.Q.shape[0]
        Pn, Qn, Pw = _f32(P_nrm), _f32(Q_nrm), _f32(P_w)
        self.h = self.L.ref_create(_fp(self.P), _fp(Pn), _fp(Pw), self.nP, _fp(self.Q), _fp(Qn), self.nQ)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.ref_destroy(self.h)
            self.h = None

    def normals(self, which):
        n = self.nQ if which else self.nP
        out = np.zeros((n, 3), dtype=np.float32)
        self.L.ref_get_normals(self.h, which, _fp(out))
        return out

    def kd_query(self, q, sqdist):
        q = _f32(q)
        return self.L.ref_kd_query(self.h, _fp(q), C.c_float(sqdist))

    def verify(self, T16, delta, best_lcp=0.0, early_out=False):
        T16 = _f32(T16)
        good = C.c_int(0)
        hits = np.full(self.nQ, -2, dtype=np.int32)
        s = self.L.ref_verify(self.h, _fp(T16), C.c_float(delta), C.c_float(best_lcp),
                              int(early_out), C.byref(good), _ip(hits))
        return float(np.float32(s)), good.value, hits

    def weighted_verify(self, T16, delta):
        T16 = _f32(T16)
        reg = np.zeros(max(self.nQ, 1), dtype=np.int32)
        n = C.c_int(0)
        s = self.L.ref_weighted_verify(self.h, _fp(T16), C.c_float(delta), _ip(reg), C.byref(n))
        return float(np.float32(s)), reg[: n.value].copy()
