"""Host-side mirror of the reference's scoring surface, over the C ABI (include/pgp.h).

Names follow the reference (S4/algorithms/match4pcsBase.{h,cc}): a `LcpScorer` plays the part of
the per-call `MatchSuper4PCS` object between `init()` and the end of `Perform_N_steps()`:

    init(P, Q_validation)            -> LcpScorer.init(...)            base.cc:216-345
    Verify(mat)                      -> LcpScorer.Verify(mat)          base.cc:1699-1731
    WeightedVerify(mat, registered)  -> LcpScorer.WeightedVerify(mat)  base.cc:1733-1766
    verification loop                -> LcpScorer.score(transforms)    base.cc:1885-1901

All computation happens in libpgp.so's HIP kernels; this file only marshals arrays.
"""
from __future__ import annotations

import ctypes as C
import numpy as np

from . import _lib
from ._lib import PGP_MODE_PLAIN, PGP_MODE_WEIGHTED

_f = C.POINTER(C.c_float)
_i = C.POINTER(C.c_int)


def _fp(a):
    return None if a is None else a.ctypes.data_as(_f)


def _f32(a, cols=None):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.float32)
    if cols is not None:
        a = a.reshape(-1, cols)
    return a


class LcpScorer:
    """One (scene, model) pair on one GPU."""

    def __init__(self, device: int = -1):
        self._lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self._lib.pgp_create(C.byref(h), int(device)))
        self._h = h
        self.nQ = 0
        self.nP = 0
        self.delta = None

    @classmethod
    def borrowed(cls, handle):
        """A view of a context somebody else owns (a member of a device group: MultiGpuScorer.member): never destroyed here."""
        self = cls.__new__(cls)
        self._lib = _lib.load()
        self._h = C.c_void_p(handle)
        self._borrowed = True
        self.nQ = 0
        self.nP = 0
        self.delta = None
        return self

    def close(self):
        if getattr(self, "_h", None):
            if not getattr(self, "_borrowed", False):
                self._lib.pgp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- init() -----------------------------------------------------------------------------
    @staticmethod
    def center(P_xyz, Qs_xyz, Qv_xyz):
        """base.cc:242-268: returns centred copies and the two centroids."""
        lib = _lib.load()
        P, Qs, Qv = (np.array(_f32(x, 3), copy=True) for x in (P_xyz, Qs_xyz, Qv_xyz))
        cP, cQ = np.zeros(3, np.float32), np.zeros(3, np.float32)
        _lib.check(lib.pgp_center(_fp(P), len(P), _fp(Qs), len(Qs), _fp(Qv), len(Qv), _fp(cP), _fp(cQ)))
        return P, Qs, Qv, cP, cQ

    @staticmethod
    def image_rows_needed(P_xyz, centroid_P, K, rows, cols):
        """The image rows weights_from_image reads for these points: (row_min, row_max), (-1, -1) when no point
        falls inside a rows x cols image (what lets the drop-in stop decoding the probability PNG early)."""
        P = _f32(P_xyz, 3)
        c = _f32(centroid_P).reshape(3)
        Kf = _f32(K).reshape(9)
        lo, hi = C.c_int(0), C.c_int(0)
        _lib.check(_lib.load().pgp_image_rows_needed(_fp(P), len(P), _fp(c), _fp(Kf), int(rows), int(cols),
                                                     C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    @staticmethod
    def weights_from_image(P_centred, centroid_P, K, img_u16):
        """base.cc:317-340: per-point weight = probability image (u16 / 10000) at the projection."""
        P = _f32(P_centred, 3)
        cP, K = _f32(centroid_P).reshape(3), _f32(K).reshape(9)
        img = np.ascontiguousarray(img_u16, np.uint16)
        out = np.zeros(len(P), np.float32)
        _lib.check(_lib.load().pgp_weights_from_image(
            _fp(P), len(P), _fp(cP), _fp(K), img.ctypes.data_as(C.POINTER(C.c_ushort)),
            img.shape[0], img.shape[1], _fp(out)))
        return out

    def set_scene(self, xyz, nrm=None, weight=None, delta=0.005):
        xyz, nrm, weight = _f32(xyz, 3), _f32(nrm, 3), _f32(weight)
        if nrm is not None and len(nrm) != len(xyz):
            raise ValueError("normals / points size mismatch")
        if weight is not None and len(weight) != len(xyz):
            raise ValueError("weights / points size mismatch")
        _lib.check(self._lib.pgp_set_scene(self._h, _fp(xyz), _fp(nrm), _fp(weight), len(xyz),
                                           C.c_float(delta)))
        self.nP, self.delta = len(xyz), float(delta)

    def set_model(self, xyz, nrm=None):
        xyz, nrm = _f32(xyz, 3), _f32(nrm, 3)
        if nrm is not None and len(nrm) != len(xyz):
            raise ValueError("normals / points size mismatch")
        _lib.check(self._lib.pgp_set_model(self._h, _fp(xyz), _fp(nrm), len(xyz)))
        self.nQ = len(xyz)

    def init(self, P_xyz, P_nrm, P_w, Q_xyz, Q_nrm, delta=0.005):
        self.set_scene(P_xyz, P_nrm, P_w, delta)
        self.set_model(Q_xyz, Q_nrm)

    # ---- rigid fit from congruent pairs (base.cc:1411-1488) ----------------------------------------
    def set_search_model(self, xyz):
        xyz = _f32(xyz, 3)
        _lib.check(self._lib.pgp_set_search_model(self._h, _fp(xyz), len(xyz)))
        self.nQs = len(xyz)

    def rigid_from_congruent(self, base_ids, quad_ids, centroid_P, centroid_Q):
        """base_ids, quad_ids: (n,4) int.  Returns (T (n,16) f32, pose (n,16) f64, status (n,), rms (n,))
        for every pair; keep status == 1 to get the reference's allTransforms / allPose lists."""
        b = np.ascontiguousarray(base_ids, np.int32).reshape(-1, 4)
        q = np.ascontiguousarray(quad_ids, np.int32).reshape(-1, 4)
        assert len(b) == len(q)
        n = len(b)
        cP, cQ = _f32(centroid_P).reshape(3), _f32(centroid_Q).reshape(3)
        T = np.zeros((n, 16), np.float32)
        pose = np.zeros((n, 16), np.float64)
        status = np.zeros(n, np.int32)
        rms = np.zeros(n, np.float32)
        _lib.check(self._lib.pgp_rigid_from_congruent(
            self._h, b.ctypes.data_as(_i), q.ctypes.data_as(_i), n, _fp(cP), _fp(cQ), _fp(T),
            pose.ctypes.data_as(C.POINTER(C.c_double)), status.ctypes.data_as(_i), _fp(rms)))
        return T, pose, status, rms

    # ---- congruent-set extraction (super4pcs.cc:78-236) ---------------------------------------------
    def extract_pairs(self, pair_distance, eps, cap=None):
        """(n,2) int32 ordered pairs of search-model ids at distance pair_distance +- eps."""
        n = C.c_int(0)
        if cap is None:
            _lib.check(self._lib.pgp_extract_pairs(self._h, C.c_float(pair_distance), C.c_float(eps), None, 0,
                                                   C.byref(n)))
            cap = n.value
        out = np.zeros((max(cap, 1), 2), np.int32)
        _lib.check(self._lib.pgp_extract_pairs(self._h, C.c_float(pair_distance), C.c_float(eps),
                                               out.ctypes.data_as(_i), int(cap), C.byref(n)))
        return out[: min(n.value, cap)].copy()

    def find_congruent(self, base, invariant1, invariant2, threshold, P_pairs, Q_pairs, cap=None):
        """(n,4) int32 congruent quadrilaterals in the reference's order."""
        base = _f32(base).reshape(12)
        Pp = np.ascontiguousarray(P_pairs, np.int32).reshape(-1, 2)
        Qp = np.ascontiguousarray(Q_pairs, np.int32).reshape(-1, 2)
        n = C.c_int(0)
        args = (self._h, _fp(base), C.c_float(invariant1), C.c_float(invariant2), C.c_float(threshold),
                Pp.ctypes.data_as(_i), len(Pp), Qp.ctypes.data_as(_i), len(Qp))
        if cap is None:
            _lib.check(self._lib.pgp_find_congruent(*args, None, 0, C.byref(n)))
            cap = n.value
        out = np.zeros((max(cap, 1), 4), np.int32)
        _lib.check(self._lib.pgp_find_congruent(*args, out.ctypes.data_as(_i), int(cap), C.byref(n)))
        return out[: min(n.value, cap)].copy()

    # ---- base selection (base.cc:600-792) -------------------------------------------------------------
    def set_ppf_map(self, keys, counts=None, pairs=None):
        """keys (n,4) int: the model's discretised pair features; counts (n,) / pairs (sum,2): their pair lists."""
        keys = np.ascontiguousarray(keys, np.int32).reshape(-1, 4)
        c = None if counts is None else np.ascontiguousarray(counts, np.int32)
        p = None if pairs is None else np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
        _lib.check(self._lib.pgp_set_ppf_map(self._h, keys.ctypes.data_as(_i),
                                             None if c is None else c.ctypes.data_as(_i),
                                             None if p is None else p.ctypes.data_as(_i), len(keys)))

    def select_bases(self, u, rows=False):
        """u (n,4) float64 uniforms in [0,1) -> (ids (n,4), invariants (n,2), status (n,)); rows=True: also the pair-feature
        table rows (n,2) of every base's two edges (pgp_select_bases_rows)."""
        u = np.ascontiguousarray(u, np.float64).reshape(-1, 4)
        n = len(u)
        ids = np.zeros((max(n, 1), 4), np.int32)
        inv = np.zeros((max(n, 1), 2), np.float32)
        st = np.zeros(max(n, 1), np.int32)
        if rows:
            rw = np.zeros((max(n, 1), 2), np.int32)
            _lib.check(self._lib.pgp_select_bases_rows(self._h, u.ctypes.data_as(C.POINTER(C.c_double)), n,
                                                       ids.ctypes.data_as(_i), _fp(inv), st.ctypes.data_as(_i),
                                                       rw.ctypes.data_as(_i)))
            return ids[:n], inv[:n], st[:n], rw[:n]
        _lib.check(self._lib.pgp_select_bases(self._h, u.ctypes.data_as(C.POINTER(C.c_double)), n,
                                              ids.ctypes.data_as(_i), _fp(inv), st.ctypes.data_as(_i)))
        return ids[:n], inv[:n], st[:n]

    def select_bases_begin(self, u):
        """pgp_select_bases_rows_begin: queue the selection and return; select_bases_end() collects (ids, inv, status, rows)."""
        u = np.ascontiguousarray(u, np.float64).reshape(-1, 4)
        self._sel_n = len(u)
        _lib.check(self._lib.pgp_select_bases_rows_begin(self._h, u.ctypes.data_as(C.POINTER(C.c_double)), len(u)))

    def select_bases_end(self):
        n = self._sel_n
        ids, inv = np.zeros((n, 4), np.int32), np.zeros((n, 2), np.float32)
        status, rows = np.zeros(n, np.int32), np.zeros((n, 2), np.int32)
        _lib.check(self._lib.pgp_select_bases_rows_end(self._h, ids.ctypes.data_as(_i), _fp(inv), status.ctypes.data_as(_i), rows.ctypes.data_as(_i)))
        return ids, inv, status, rows

    def ppf_features(self, pairs):
        """pairs (m,2) scene ids -> (features (m,4), table row (m,) or -1)."""
        pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
        m = len(pairs)
        f = np.zeros((max(m, 1), 4), np.int32)
        rows = np.zeros(max(m, 1), np.int32)
        _lib.check(self._lib.pgp_ppf_features(self._h, pairs.ctypes.data_as(_i), m, f.ctypes.data_as(_i),
                                              rows.ctypes.data_as(_i)))
        return f[:m], rows[:m]

    def stocs_stage_weights(self, stage, cur, base1, base2=-1, base3=-1):
        """One weighting loop of SelectQuadrilateralStoCS -> (cur_out, sum, present)."""
        cur = np.array(_f32(cur), copy=True)
        s, p = C.c_float(0), C.c_int(0)
        _lib.check(self._lib.pgp_stocs_stage_weights(self._h, int(stage), int(base1), int(base2), int(base3),
                                                     _fp(cur), C.byref(s), C.byref(p)))
        return cur, float(np.float32(s.value)), bool(p.value)

    def base_invariants(self, ids):
        """TryQuadrilateral for (m,4) scene ids -> (reordered ids, invariants (m,2), ok (m,))."""
        ids = np.array(np.ascontiguousarray(ids, np.int32).reshape(-1, 4), copy=True)
        m = len(ids)
        inv = np.zeros((max(m, 1), 2), np.float32)
        ok = np.zeros(max(m, 1), np.int32)
        _lib.check(self._lib.pgp_base_invariants(self._h, ids.ctypes.data_as(_i), m, _fp(inv), ok.ctypes.data_as(_i)))
        return ids, inv[:m], ok[:m]

    def find_congruent_batch(self, base_ids, base_xyz, invariants, threshold, rows=None):
        """All bases of an object at once (pairs from the device PPF table): returns the quad counts (nb,).  rows (nb,2):
        the table rows of the bases' edges when the caller holds them (select_bases(rows=True)): pgp_find_congruent_batch_rows."""
        b = np.ascontiguousarray(base_ids, np.int32).reshape(-1, 4)
        x = _f32(base_xyz).reshape(-1, 12)
        v = _f32(invariants).reshape(-1, 2)
        n = np.zeros(max(len(b), 1), np.int32)
        if rows is not None:
            rw = np.ascontiguousarray(rows, np.int32).reshape(-1, 2)
            assert len(rw) == len(b)
            _lib.check(self._lib.pgp_find_congruent_batch_rows(self._h, b.ctypes.data_as(_i), _fp(x), _fp(v), rw.ctypes.data_as(_i),
                                                           len(b), C.c_float(threshold), n.ctypes.data_as(_i)))
            return n[: len(b)]
        _lib.check(self._lib.pgp_find_congruent_batch(self._h, b.ctypes.data_as(_i), _fp(x), _fp(v), len(b),
                                                      C.c_float(threshold), n.ctypes.data_as(_i)))
        return n[: len(b)]

    def congruent_batch_fit_score_list(self, picks, base_ids, centroid_P, centroid_Q, mode=PGP_MODE_WEIGHTED, gate_deg=30.0, list_cap=256):
        """pgp_congruent_batch_fit_score_list: fits + verification + the running-best walk + the kept poses + the best pose and
        its registered points in one call.  Returns a dict: n_list, index, score, T, pose (the first min(n_list, list_cap)
        records), n_pushed, best_index, best_score, best_T, best_pose, registered."""
        pk = np.ascontiguousarray(picks, np.int32).reshape(-1, 2)
        b = np.ascontiguousarray(base_ids, np.int32).reshape(-1, 4)
        cP, cQ = _f32(centroid_P).reshape(3), _f32(centroid_Q).reshape(3)
        cap = int(list_cap)
        li = np.zeros(max(cap, 1), np.int32)
        ls = np.zeros(max(cap, 1), np.float32)
        lT = np.zeros((max(cap, 1), 16), np.float32)
        lp = np.zeros((max(cap, 1), 16), np.float64)
        bT, bp = np.zeros(16, np.float32), np.zeros(16, np.float64)
        reg = np.zeros(max(self.nQ, 1), np.int32)
        n_list, n_pushed, best, n_reg = C.c_int(0), C.c_int(0), C.c_int(-1), C.c_int(0)
        bs = C.c_float(0)
        _lib.check(self._lib.pgp_congruent_batch_fit_score_list(
            self._h, pk.ctypes.data_as(_i), b.ctypes.data_as(_i), len(pk), _fp(cP), _fp(cQ), int(mode), C.c_float(gate_deg), cap,
            C.byref(n_list), li.ctypes.data_as(_i), _fp(ls), _fp(lT), lp.ctypes.data_as(C.POINTER(C.c_double)), C.byref(n_pushed),
            C.byref(best), C.byref(bs), _fp(bT), bp.ctypes.data_as(C.POINTER(C.c_double)), reg.ctypes.data_as(_i), C.byref(n_reg)))
        k = min(n_list.value, cap)
        return dict(n_list=n_list.value, index=li[:k].copy(), score=ls[:k].copy(), T=lT[:k].copy(), pose=lp[:k].copy(),
                    n_pushed=n_pushed.value, best_index=best.value, best_score=float(np.float32(bs.value)), best_T=bT, best_pose=bp,
                    registered=reg[: n_reg.value].copy())

    @staticmethod
    def sample_quads(seed, n_quads, max_per_base=100):
        """pgp_sample_quads: the picks (base, quad) the device draws for these quad counts (host statement of the same draw)."""
        nq = np.ascontiguousarray(n_quads, np.int32)
        out = np.zeros((max(len(nq) * int(max_per_base), 1), 2), np.int32)
        n = C.c_int(0)
        _lib.check(_lib.load().pgp_sample_quads(C.c_ulonglong(int(seed)), nq.ctypes.data_as(_i), len(nq), int(max_per_base),
                                                out.ctypes.data_as(_i), C.byref(n)))
        return out[: n.value].copy()

    def congruent_batch_sample_fit_score_list(self, seed, base_ids, centroid_P, centroid_Q, max_per_base=100, mode=PGP_MODE_WEIGHTED,
                                              gate_deg=30.0, list_cap=256):
        """pgp_congruent_batch_sample_fit_score_list: congruent_batch_fit_score_list with the quads drawn on the device.
        The dict of that call + `picks` (what was drawn)."""
        b = np.ascontiguousarray(base_ids, np.int32).reshape(-1, 4)
        cP, cQ = _f32(centroid_P).reshape(3), _f32(centroid_Q).reshape(3)
        cap = int(list_cap)
        li = np.zeros(max(cap, 1), np.int32)
        ls = np.zeros(max(cap, 1), np.float32)
        lT = np.zeros((max(cap, 1), 16), np.float32)
        lp = np.zeros((max(cap, 1), 16), np.float64)
        bT, bp = np.zeros(16, np.float32), np.zeros(16, np.float64)
        reg = np.zeros(max(self.nQ, 1), np.int32)
        pk = np.zeros((max(len(b) * int(max_per_base), 1), 2), np.int32)
        n_list, n_pushed, best, n_reg, n_pk = C.c_int(0), C.c_int(0), C.c_int(-1), C.c_int(0), C.c_int(0)
        bs = C.c_float(0)
        _lib.check(self._lib.pgp_congruent_batch_sample_fit_score_list(
            self._h, C.c_ulonglong(int(seed)), int(max_per_base), b.ctypes.data_as(_i), _fp(cP), _fp(cQ), int(mode), C.c_float(gate_deg), cap,
            C.byref(n_list), li.ctypes.data_as(_i), _fp(ls), _fp(lT), lp.ctypes.data_as(C.POINTER(C.c_double)), C.byref(n_pushed),
            C.byref(best), C.byref(bs), _fp(bT), bp.ctypes.data_as(C.POINTER(C.c_double)), reg.ctypes.data_as(_i), C.byref(n_reg),
            pk.ctypes.data_as(_i), C.byref(n_pk)))
        k = min(n_list.value, cap)
        return dict(n_list=n_list.value, index=li[:k].copy(), score=ls[:k].copy(), T=lT[:k].copy(), pose=lp[:k].copy(),
                    n_pushed=n_pushed.value, best_index=best.value, best_score=float(np.float32(bs.value)), best_T=bT, best_pose=bp,
                    registered=reg[: n_reg.value].copy(), picks=pk[: n_pk.value].copy())

    def congruent_batch_quads(self, picks):
        pk = np.ascontiguousarray(picks, np.int32).reshape(-1, 2)
        out = np.zeros((max(len(pk), 1), 4), np.int32)
        _lib.check(self._lib.pgp_congruent_batch_quads(self._h, pk.ctypes.data_as(_i), len(pk), out.ctypes.data_as(_i)))
        return out[: len(pk)]

    def congruent_batch_fit(self, picks, base_ids, centroid_P, centroid_Q):
        pk = np.ascontiguousarray(picks, np.int32).reshape(-1, 2)
        b = np.ascontiguousarray(base_ids, np.int32).reshape(-1, 4)
        n = len(pk)
        cP, cQ = _f32(centroid_P).reshape(3), _f32(centroid_Q).reshape(3)
        T = np.zeros((max(n, 1), 16), np.float32)
        pose = np.zeros((max(n, 1), 16), np.float64)
        status = np.zeros(max(n, 1), np.int32)
        rms = np.zeros(max(n, 1), np.float32)
        _lib.check(self._lib.pgp_congruent_batch_fit(
            self._h, pk.ctypes.data_as(_i), b.ctypes.data_as(_i), n, _fp(cP), _fp(cQ), _fp(T),
            pose.ctypes.data_as(C.POINTER(C.c_double)), status.ctypes.data_as(_i), _fp(rms)))
        return T[:n], pose[:n], status[:n], rms[:n]

    # ---- segment pre-processing (ObjectPoseCandidateSet.cpp:28-51) -----------------------------------
    def radius_outlier_filter(self, xyz, nrm=None, radius=0.03, min_neighbors=10):
        """Returns (keep mask (n,) bool, flipped + re-normalised normals (n,3) or None)."""
        xyz, nrm = _f32(xyz, 3), _f32(nrm, 3)
        n = len(xyz)
        keep = np.zeros(max(n, 1), np.uint8)
        nout = np.zeros((max(n, 1), 3), np.float32) if nrm is not None else None
        kept = C.c_int(0)
        _lib.check(self._lib.pgp_radius_outlier_filter(
            self._h, _fp(xyz), _fp(nrm), n, C.c_float(radius), int(min_neighbors),
            keep.ctypes.data_as(C.POINTER(C.c_ubyte)), _fp(nout), C.byref(kept)))
        assert kept.value == int(keep[:n].sum())
        return keep[:n].astype(bool), (nout[:n] if nout is not None else None)

    def voxel_grid(self, xyz, leaf=0.01):
        """pcl::VoxelGrid(leaf) centroids in ascending voxel index (Segmentation.cpp:234-237)."""
        xyz = _f32(xyz, 3)
        n = len(xyz)
        out = np.zeros((max(n, 1), 3), np.float32)
        m = C.c_int(0)
        _lib.check(self._lib.pgp_voxel_grid(self._h, _fp(xyz), n, C.c_float(leaf), _fp(out), n, C.byref(m)))
        return out[: m.value].copy()

    def mls_normals(self, xyz, radius=0.02):
        """pcl::MovingLeastSquares (polynomial order 2, normals, no upsampling; Segmentation.cpp:239-246):
        (smoothed xyz, un-normalised normals, curvature, input index) of the points with >= 3 neighbours."""
        xyz = _f32(xyz, 3)
        n = len(xyz)
        cap = max(n, 1)
        ox, on = np.zeros((cap, 3), np.float32), np.zeros((cap, 3), np.float32)
        oc, oi = np.zeros(cap, np.float32), np.zeros(cap, np.int32)
        m = C.c_int(0)
        _lib.check(self._lib.pgp_mls_normals(self._h, _fp(xyz), n, C.c_float(radius), _fp(ox), _fp(on), _fp(oc),
                                             oi.ctypes.data_as(_i), cap, C.byref(m)))
        return ox[: m.value].copy(), on[: m.value].copy(), oc[: m.value].copy(), oi[: m.value].copy()

    def mls_normals_device(self, d_xyz, n, radius, d_out_xyz, d_out_nrm=None, d_out_curv=None, d_out_index=None):
        import torch
        m = C.c_int(0)
        st = torch.cuda.current_stream(d_xyz.device).cuda_stream
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        _lib.check(self._lib.pgp_mls_normals_device(self._h, p(d_xyz), int(n), C.c_float(radius), p(d_out_xyz), p(d_out_nrm),
                                                    p(d_out_curv), p(d_out_index), int(d_out_xyz.shape[0]), C.byref(m),
                                                    C.c_void_p(st)))
        return m.value

    def pose_hausdorff(self, hull_xyz, T, pairs):
        """c_dist_pose / c_dist_pose_mean (base.cc:1616-1655) for (m,2) index pairs into T (n,16)."""
        hull, T = _f32(hull_xyz, 3), _f32(T, 16)
        pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
        m = len(pairs)
        dmax, dsum = np.zeros(max(m, 1), np.float32), np.zeros(max(m, 1), np.float32)
        _lib.check(self._lib.pgp_pose_hausdorff(self._h, _fp(hull), len(hull), _fp(T), len(T), pairs.ctypes.data_as(_i), m,
                                                _fp(dmax), _fp(dsum)))
        return dmax[:m], dsum[:m]

    # device-resident chain: depth image -> cloud -> voxel grid -> scene index (torch tensors carry the memory)
    def backproject_depth_device(self, d_image, K, d_mask, d_xyz_out, z_min=0.1, z_max=2.0):
        import torch
        rows, cols = d_image.shape
        assert d_image.is_cuda and d_image.is_contiguous() and d_image.dtype in (torch.uint16, torch.int16, torch.float32)
        raw16 = int(d_image.dtype != torch.float32)
        K9 = np.ascontiguousarray(K, np.float32).reshape(9)
        n = C.c_int(0)
        st = torch.cuda.current_stream(d_image.device).cuda_stream
        _lib.check(self._lib.pgp_backproject_depth_device(
            self._h, C.c_void_p(d_image.data_ptr()), raw16, C.c_void_p(d_mask.data_ptr()) if d_mask is not None else None,
            rows, cols, _fp(K9), C.c_double(z_min), C.c_double(z_max), C.c_void_p(d_xyz_out.data_ptr()),
            int(d_xyz_out.shape[0]), C.byref(n), C.c_void_p(st)))
        return n.value

    def voxel_grid_device(self, d_xyz, n, leaf, d_out):
        import torch
        m = C.c_int(0)
        st = torch.cuda.current_stream(d_xyz.device).cuda_stream
        _lib.check(self._lib.pgp_voxel_grid_device(self._h, C.c_void_p(d_xyz.data_ptr()), int(n), C.c_float(leaf),
                                                   C.c_void_p(d_out.data_ptr()), int(d_out.shape[0]), C.byref(m),
                                                   C.c_void_p(st)))
        return m.value

    def set_scene_device(self, d_xyz, n, d_nrm=None, d_weight=None, delta=0.005):
        import torch
        st = torch.cuda.current_stream(d_xyz.device).cuda_stream
        _lib.check(self._lib.pgp_set_scene_device(
            self._h, C.c_void_p(d_xyz.data_ptr()), C.c_void_p(d_nrm.data_ptr()) if d_nrm is not None else None,
            C.c_void_p(d_weight.data_ptr()) if d_weight is not None else None, int(n), C.c_float(delta), C.c_void_p(st)))
        self.nP, self.delta = int(n), float(delta)

    def cluster_poses_device(self, d_T, d_scores, best_score, d_rep, d_assign, sym_deg=(0, 0, 0), accept_fraction=0.5,
                             rot_thresh_deg=10.0, trans_thresh=0.02):
        import torch
        sym = np.ascontiguousarray(sym_deg, np.float32)
        prm = _lib.ClusterParams(float(accept_fraction), float(rot_thresh_deg), float(trans_thresh))
        n_rep = C.c_int(0)
        st = torch.cuda.current_stream(d_T.device).cuda_stream
        _lib.check(self._lib.pgp_cluster_poses_device(
            self._h, C.c_void_p(d_T.data_ptr()), C.c_void_p(d_scores.data_ptr()), int(d_T.shape[0]), C.c_float(best_score),
            _fp(sym), C.byref(prm), C.c_void_p(d_rep.data_ptr()), C.c_void_p(d_assign.data_ptr()), C.byref(n_rep),
            C.c_void_p(st)))
        return n_rep.value

    # ---- MCTS leaf cost (UCTState::computeCost) ---------------------------------------------------------
    def backproject_depth(self, image, K, mask=None, z_min=0.1, z_max=2.0):
        """image (rows, cols): uint16 raw PNG samples or float32 metres; K 3x3; mask (rows, cols) or None
        -> (n, 3) float32 camera-frame cloud in the reference's scan order."""
        image = np.ascontiguousarray(image)
        assert image.dtype in (np.uint16, np.float32) and image.ndim == 2
        rows, cols = image.shape
        K9 = np.ascontiguousarray(K, np.float32).reshape(9)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        out = np.zeros((max(rows * cols, 1), 3), np.float32)
        n = C.c_int(0)
        _lib.check(self._lib.pgp_backproject_depth(
            self._h, image.ctypes.data_as(C.c_void_p), int(image.dtype == np.uint16),
            None if m is None else m.ctypes.data_as(C.POINTER(C.c_ubyte)), rows, cols, _fp(K9),
            C.c_double(z_min), C.c_double(z_max), _fp(out), rows * cols, C.byref(n)))
        return out[: n.value].copy()

    def depth_cost(self, observed, rendered, threshold=0.01):
        """observed (rows,cols) f32, rendered (n,rows,cols) f32 -> (render_score (n,), counts (n,3))."""
        obs = np.ascontiguousarray(observed, np.float32)
        ren = np.ascontiguousarray(rendered, np.float32).reshape(-1, *obs.shape)
        n = len(ren)
        score = np.zeros(n, np.float32)
        counts = np.zeros((n, 3), np.int32)
        _lib.check(self._lib.pgp_depth_cost(self._h, _fp(obs), _fp(ren), n, obs.shape[0], obs.shape[1],
                                            C.c_float(threshold), _fp(score), counts.ctypes.data_as(_i)))
        return score, counts

    def unexplained_segment(self, seg_xyz, models, poses, radius=0.008):
        """UCTState::performTrICP's pre-filter (UCTState.cpp:142-174): models = list of (m_k,3) clouds of the objects
        already placed, poses = (K,16) column-major model -> segment frame.  Returns keep (n,) bool."""
        seg = _f32(seg_xyz, 3)
        K = len(models)
        off = np.zeros(K + 1, np.int32)
        for k, m in enumerate(models):
            off[k + 1] = off[k] + len(m)
        allm = _f32(np.concatenate([np.asarray(m, np.float32).reshape(-1, 3) for m in models]) if K else np.zeros((0, 3)), 3)
        T = _f32(np.asarray(poses, np.float32).reshape(-1, 16) if K else np.zeros((0, 16)), 16)
        keep = np.zeros(max(len(seg), 1), np.uint8)
        n_kept = C.c_int(0)
        _lib.check(self._lib.pgp_unexplained_segment(self._h, _fp(seg), len(seg), _fp(allm), off.ctypes.data_as(_i), _fp(T), K,
                                                     C.c_float(radius), keep.ctypes.data_as(C.c_char_p), C.byref(n_kept)))
        keep = keep[:len(seg)].astype(bool)
        assert int(keep.sum()) == n_kept.value
        return keep

    def depth_cost_device(self, d_observed, d_rendered, threshold=0.01, d_counts=None, d_scores=None, stream=None):
        """cuda float32 tensors: observed (rows,cols), rendered (n,rows,cols) -> (d_counts (n,3) int32, d_scores (n,))
        on the device; enqueued, no synchronisation."""
        import torch
        n, rows, cols = d_rendered.shape
        assert d_observed.is_cuda and d_rendered.is_cuda and d_observed.is_contiguous() and d_rendered.is_contiguous()
        if d_counts is None:
            d_counts = torch.empty((n, 3), dtype=torch.int32, device=d_rendered.device)
        if d_scores is None:
            d_scores = torch.empty(n, dtype=torch.float32, device=d_rendered.device)
        st = stream if isinstance(stream, int) else (stream or torch.cuda.current_stream(d_rendered.device)).cuda_stream
        _lib.check(self._lib.pgp_depth_cost_device(self._h, C.c_void_p(d_observed.data_ptr()), C.c_void_p(d_rendered.data_ptr()),
                                                   int(n), int(rows), int(cols), C.c_float(threshold),
                                                   C.c_void_p(d_counts.data_ptr()), C.c_void_p(d_scores.data_ptr()),
                                                   C.c_void_p(st)))
        return d_counts, d_scores

    @staticmethod
    def camera(K, rows, cols, z_near=0.1, z_max=1.0):
        """pgp_camera from a 3x3 intrinsic matrix; z_max 1.0 = renderScene.cpp:69."""
        K = np.asarray(K, np.float32)
        return _lib.Camera(int(rows), int(cols), float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]),
                           float(z_near), float(z_max))

    def render_depth(self, vertices, triangles, T, cam, parent=None):
        """Host arrays: vertices (n_vert,3), triangles (n_tri,3) int32 or None (point splat), T (n,16) col-major
        object -> camera, parent (rows,cols) or None -> depth (n,rows,cols) float32."""
        v = _f32(vertices, 3)
        T = _f32(T, 16)
        tri = None if triangles is None else np.ascontiguousarray(triangles, np.int32).reshape(-1, 3)
        par = None if parent is None else np.ascontiguousarray(parent, np.float32)
        out = np.zeros((len(T), cam.rows, cam.cols), np.float32)
        _lib.check(self._lib.pgp_render_depth(self._h, _fp(v), len(v), None if tri is None else tri.ctypes.data_as(_i),
                                              0 if tri is None else len(tri), _fp(T), len(T), C.byref(cam),
                                              None if par is None else _fp(par), _fp(out)))
        return out

    def render_depth_device(self, d_vertices, d_triangles, d_T, cam, d_parent=None, d_depth=None, stream=None):
        """cuda tensors: vertices (n_vert,3|4) float32, triangles (n_tri,3) int32 or None, T (n,16), parent None |
        (rows,cols) shared | (n,rows,cols) one per image -> d_depth (n,rows,cols); enqueued, no synchronisation."""
        import torch
        n = int(d_T.shape[0])
        if d_depth is None:
            d_depth = torch.empty((n, cam.rows, cam.cols), dtype=torch.float32, device=d_T.device)
        stride = 0
        if d_parent is not None and d_parent.dim() == 3:
            stride = cam.rows * cam.cols
        st = stream if isinstance(stream, int) else (stream or torch.cuda.current_stream(d_T.device)).cuda_stream
        _lib.check(self._lib.pgp_render_depth_device(
            self._h, C.c_void_p(d_vertices.data_ptr()), int(d_vertices.shape[1]), int(d_vertices.shape[0]),
            C.c_void_p(0 if d_triangles is None else d_triangles.data_ptr()), 0 if d_triangles is None else int(d_triangles.shape[0]),
            C.c_void_p(d_T.data_ptr()), n, C.byref(cam), C.c_void_p(0 if d_parent is None else d_parent.data_ptr()),
            C.c_size_t(stride), C.c_void_p(d_depth.data_ptr()), C.c_void_p(st)))
        return d_depth

    # ---- hypothesis clustering (HypothesisSelection::greedyClustering) ---------------------------
    def cluster_poses(self, T, scores, best_score=None, sym_deg=(0, 0, 0), accept_fraction=0.5,
                      rot_thresh_deg=10.0, trans_thresh=0.02):
        """T (n,16) col-major, scores (n,) -> (representative ids in clusteredHypothesisSet order,
        assignment (n,): id of the absorbing representative, -1 when pruned)."""
        T, scores = _f32(T, 16), np.ascontiguousarray(scores, np.float32)
        n = len(T)
        if best_score is None:
            best_score = float(scores.max()) if n else 0.0
        sym = np.ascontiguousarray(sym_deg, np.float32)
        prm = _lib.ClusterParams(float(accept_fraction), float(rot_thresh_deg), float(trans_thresh))
        rep = np.zeros(max(n, 1), np.int32)
        assign = np.zeros(max(n, 1), np.int32)
        n_rep = C.c_int(0)
        _lib.check(self._lib.pgp_cluster_poses(self._h, _fp(T), _fp(scores), n, C.c_float(best_score), _fp(sym),
                                               C.byref(prm), rep.ctypes.data_as(_i), n, C.byref(n_rep),
                                               assign.ctypes.data_as(_i)))
        return rep[: n_rep.value].copy(), assign[:n].copy()

    def pose_error(self, test, gt, sym_deg=(0, 0, 0)):
        """utilities::getPoseError for n pairs -> (mean rotation error in degrees, translation error)."""
        test, gt = _f32(test, 16), _f32(gt, 16)
        n = len(test)
        sym = np.ascontiguousarray(sym_deg, np.float32)
        rot, trans = np.zeros(n, np.float32), np.zeros(n, np.float32)
        _lib.check(self._lib.pgp_pose_error(self._h, _fp(test), _fp(gt), n, _fp(sym), _fp(rot), _fp(trans)))
        return rot, trans

    # ---- ICP refinement (UCTState::performTrICP / utilities::performICP inner loop) --------------
    def icp_refine(self, src_xyz, tgt_xyz, T, trim=1.0, max_iterations=100, max_corr_dist=0.0,
                   energy_ratio=1.0):
        """T: (n,16) column-major guesses (source -> target frame).  Returns (T_refined, energy, iters)."""
        src, tgt = _f32(src_xyz, 3), _f32(tgt_xyz, 3)
        T = np.array(_f32(T, 16), copy=True)
        n = len(T)
        prm = _lib.IcpParams(int(max_iterations), float(trim), float(max_corr_dist), float(energy_ratio))
        energy = np.zeros(n, np.float32)
        iters = np.zeros(n, np.int32)
        _lib.check(self._lib.pgp_icp_refine(self._h, _fp(src), len(src), _fp(tgt), len(tgt), _fp(T), n,
                                            C.byref(prm), _fp(energy), iters.ctypes.data_as(_i)))
        return T, energy, iters

    def icp_refine_ex(self, src_xyz, tgt_xyz, T, tgt_nrm=None, **opts):
        """The general form (pgp_icp_options): keyword arguments are the option fields, e.g.
        max_iterations=50, max_corr_dist=0.01, energy_ratio=0, transformation_epsilon=1e-8, absolute_mse=1e-12
        (greedy_bfs/State.cpp:139-142) or error_metric=1 with tgt_nrm (utilities.cpp:709-739)."""
        src, tgt, nrm = _f32(src_xyz, 3), _f32(tgt_xyz, 3), _f32(tgt_nrm, 3)
        T = np.array(_f32(T, 16), copy=True)
        n = len(T)
        o = _lib.IcpOptions()
        _lib.check(self._lib.pgp_icp_default_options(C.byref(o)))
        for k, v in opts.items():
            if not hasattr(o, k):
                raise TypeError(f"unknown ICP option {k}")
            setattr(o, k, v)
        energy = np.zeros(n, np.float32)
        iters = np.zeros(n, np.int32)
        _lib.check(self._lib.pgp_icp_refine_ex(self._h, _fp(src), len(src), _fp(tgt), _fp(nrm), len(tgt), _fp(T), n,
                                               C.byref(o), _fp(energy), iters.ctypes.data_as(_i)))
        return T, energy, iters

    def icp_refine_device(self, d_src4, d_tgt4, d_T, d_energy=None, d_iters=None, trim=1.0, max_iterations=100,
                          max_corr_dist=0.0, energy_ratio=1.0, target_token=None, stream=None):
        """pgp_icp_refine_device: d_src4 / d_tgt4 cuda float32 (n,4) {x,y,z,-}; d_T cuda float32 (n_poses,16), refined
        in place; d_energy float32 / d_iters int32 (n_poses,) or None.  target_token (non-zero int) vouches that the
        target is the one of the previous call with the same token (the index stays resident).  Enqueued, no sync."""
        import torch
        for x in (d_src4, d_tgt4, d_T):
            assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
        if stream is None:
            stream = torch.cuda.current_stream(d_T.device).cuda_stream
        elif hasattr(stream, "cuda_stream"):
            stream = stream.cuda_stream
        if target_token is not None:
            _lib.check(self._lib.pgp_icp_target_token(self._h, C.c_ulonglong(int(target_token))))
        prm = _lib.IcpParams(int(max_iterations), float(trim), float(max_corr_dist), float(energy_ratio))
        _lib.check(self._lib.pgp_icp_refine_device(
            self._h, C.c_void_p(d_src4.data_ptr()), int(d_src4.shape[0]), C.c_void_p(d_tgt4.data_ptr()),
            int(d_tgt4.shape[0]), C.c_void_p(d_T.data_ptr()), int(d_T.shape[0]), C.byref(prm),
            C.c_void_p(d_energy.data_ptr()) if d_energy is not None else None,
            C.c_void_p(d_iters.data_ptr()) if d_iters is not None else None, C.c_void_p(stream)))

    def select_top_device(self, d_T, d_scores, k, invert=True, d_T_out=None, d_index_out=None, d_n_out=None, stream=None):
        """pgp_select_top_device: the k best-scoring transforms (descending score, ties: lower index), rigidly inverted
        when `invert`, written to d_T_out (k,16); returns (d_T_out, d_index_out, d_n_out) -- all on the device."""
        import torch
        assert d_T.is_cuda and d_T.dtype == torch.float32 and d_T.is_contiguous()
        assert d_scores.is_cuda and d_scores.dtype == torch.float32 and d_scores.is_contiguous()
        n = int(d_T.shape[0])
        dev = d_T.device
        if d_T_out is None:
            d_T_out = torch.empty(k, 16, dtype=torch.float32, device=dev)
        if d_index_out is None:
            d_index_out = torch.empty(k, dtype=torch.int32, device=dev)
        if d_n_out is None:
            d_n_out = torch.zeros(1, dtype=torch.int32, device=dev)
        if stream is None:
            stream = torch.cuda.current_stream(dev).cuda_stream
        elif hasattr(stream, "cuda_stream"):
            stream = stream.cuda_stream
        _lib.check(self._lib.pgp_select_top_device(
            self._h, C.c_void_p(d_T.data_ptr()), C.c_void_p(d_scores.data_ptr()), n, int(k), 1 if invert else 0,
            C.c_void_p(d_T_out.data_ptr()), C.c_void_p(d_index_out.data_ptr()), C.c_void_p(d_n_out.data_ptr()),
            C.c_void_p(stream)))
        return d_T_out, d_index_out, d_n_out

    @staticmethod
    def icp_refine_multi_device(jobs, trim=1.0, max_iterations=100, max_corr_dist=0.0, energy_ratio=1.0, stream=None):
        """pgp_icp_refine_multi_device.  jobs: list of dicts {scorer, d_src4, d_tgt4, d_T, d_energy (opt), d_iters (opt),
        target_token (opt)}: every (segment, target) pair refined by ONE launch; each d_T refined in place."""
        import torch
        if not jobs:
            return
        lib = jobs[0]["scorer"]._lib
        arr = (_lib.IcpJob * len(jobs))()
        for j, q in enumerate(jobs):
            sc = q["scorer"]
            for x in (q["d_src4"], q["d_tgt4"], q["d_T"]):
                assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
            if q.get("target_token") is not None:
                _lib.check(lib.pgp_icp_target_token(sc._h, C.c_ulonglong(int(q["target_token"]))))
            e, it = q.get("d_energy"), q.get("d_iters")
            arr[j] = _lib.IcpJob(sc._h, q["d_src4"].data_ptr(), int(q["d_src4"].shape[0]), q["d_tgt4"].data_ptr(),
                                 int(q["d_tgt4"].shape[0]), q["d_T"].data_ptr(), int(q["d_T"].shape[0]),
                                 e.data_ptr() if e is not None else None, it.data_ptr() if it is not None else None)
        if stream is None:
            stream = torch.cuda.current_stream(jobs[0]["d_T"].device).cuda_stream
        elif hasattr(stream, "cuda_stream"):
            stream = stream.cuda_stream
        prm = _lib.IcpParams(int(max_iterations), float(trim), float(max_corr_dist), float(energy_ratio))
        _lib.check(lib.pgp_icp_refine_multi_device(arr, len(jobs), C.byref(prm), C.c_void_p(stream)))

    # ---- verification loop -----------------------------------------------------------------------
    def score(self, T, mode=PGP_MODE_PLAIN, gate_deg=30.0):
        """T: (n_h,16) column-major float transforms.  Returns (scores, counts, best_index, best_score)."""
        T = _f32(T, 16)
        n_h = len(T)
        scores = np.zeros(n_h, np.float32)
        counts = np.zeros(n_h, np.int32)
        bi = C.c_int(-1)
        bs = C.c_float(0)
        _lib.check(self._lib.pgp_score_lcp(self._h, _fp(T), n_h, int(mode), C.c_float(gate_deg),
                                           _fp(scores), counts.ctypes.data_as(_i), C.byref(bi), C.byref(bs)))
        return scores, counts, bi.value, float(np.float32(bs.value))

    def Verify(self, mat16):
        s, c, _, _ = self.score(np.asarray(mat16, np.float32).reshape(1, 16), PGP_MODE_PLAIN)
        return float(s[0]), int(c[0])

    def WeightedVerify(self, mat16, gate_deg=30.0):
        mat16 = _f32(mat16).reshape(16)
        s, _, _, _ = self.score(mat16.reshape(1, 16), PGP_MODE_WEIGHTED, gate_deg)
        return float(s[0]), self.registered(mat16, PGP_MODE_WEIGHTED, gate_deg)

    def registered(self, mat16, mode=PGP_MODE_WEIGHTED, gate_deg=30.0):
        mat16 = _f32(mat16).reshape(16)
        ids = np.zeros(max(self.nQ, 1), np.int32)
        n = C.c_int(0)
        _lib.check(self._lib.pgp_registered(self._h, _fp(mat16), int(mode), C.c_float(gate_deg),
                                            ids.ctypes.data_as(_i), C.byref(n)))
        return ids[: n.value].copy()

    def registered_model(self, mat16, q_xyz, q_nrm, gate_deg=30.0):
        """getRegisteredModel (base.cc:347-375): scene ids registered by another cloud, directed-angle gate."""
        mat16, q, qn = _f32(mat16).reshape(16), _f32(q_xyz, 3), _f32(q_nrm, 3)
        ids = np.zeros(max(len(q), 1), np.int32)
        n = C.c_int(0)
        _lib.check(self._lib.pgp_registered_model(self._h, _fp(mat16), _fp(q), _fp(qn), len(q), C.c_float(gate_deg),
                                                  ids.ctypes.data_as(_i), C.byref(n)))
        return ids[: n.value].copy()

    def find_congruent_4pcs(self, invariant1, invariant2, threshold, P_pairs, Q_pairs, cap=None):
        """Match4PCS::FindCongruentQuadrilaterals (4pcs.cc:61-103) -> (n,4) quads in (Q-pair, P-pair) order."""
        Pp = np.ascontiguousarray(P_pairs, np.int32).reshape(-1, 2)
        Qp = np.ascontiguousarray(Q_pairs, np.int32).reshape(-1, 2)
        n = C.c_int(0)
        args = (self._h, C.c_float(invariant1), C.c_float(invariant2), C.c_float(threshold), Pp.ctypes.data_as(_i),
                len(Pp), Qp.ctypes.data_as(_i), len(Qp))
        if cap is None:
            _lib.check(self._lib.pgp_find_congruent_4pcs(*args, None, 0, C.byref(n)))
            cap = n.value
        out = np.zeros((max(cap, 1), 4), np.int32)
        _lib.check(self._lib.pgp_find_congruent_4pcs(*args, out.ctypes.data_as(_i), int(cap), C.byref(n)))
        return out[: min(n.value, cap)].copy()

    def set_exact_ties(self, on=True):
        """Exact distance ties go to the scene point the reference's kd-tree returns (pgp_set_exact_ties); call before
        set_scene / init: the tree is built with the scene."""
        _lib.check(self._lib.pgp_set_exact_ties(self._h, int(bool(on))))

    def set_exact_records(self, on=True):
        """Weighted scoring calls also settle every near-record of the running-best walk exactly
        (pgp_set_exact_records): running_best(scores) is then the reference's list."""
        _lib.check(self._lib.pgp_set_exact_records(self._h, int(bool(on))))

    @staticmethod
    def running_best(scores):
        scores = _f32(scores)
        sel = np.zeros(max(len(scores), 1), np.int32)
        n = C.c_int(0)
        _lib.check(_lib.load().pgp_running_best(_fp(scores), len(scores), sel.ctypes.data_as(_i), C.byref(n)))
        return sel[: n.value].copy()

    # ---- device-resident path (torch tensors only carry memory + stream) --------------------------
    def set_verify_early_out(self, on=True):
        """plain-mode scores / counts as the reference's Verify returns them, with its early termination
        (base.cc:1708,1725-1727); off = every hypothesis counted completely"""
        _lib.check(self._lib.pgp_set_verify_early_out(self._h, int(bool(on))))

    def reserve(self, max_hypotheses):
        _lib.check(self._lib.pgp_reserve(self._h, int(max_hypotheses)))

    def score_device(self, d_T, d_scores, d_counts=None, d_best=None, mode=PGP_MODE_PLAIN,
                     gate_deg=30.0, stream=None):
        """d_T: cuda float32 tensor (n_h,16); d_scores: cuda float32 (n_h,); d_counts int32 (n_h,)
        or None; d_best int32 (2,) or None.  Enqueued on `stream` (torch stream or raw handle;
        default: torch's current stream).  Does not synchronise."""
        import torch
        assert d_T.is_cuda and d_T.dtype == torch.float32 and d_T.is_contiguous()
        assert d_scores.is_cuda and d_scores.dtype == torch.float32 and d_scores.is_contiguous()
        n_h = int(d_T.shape[0])
        assert d_scores.numel() >= n_h
        if d_counts is not None:
            assert d_counts.is_cuda and d_counts.dtype == torch.int32 and d_counts.numel() >= n_h
        if d_best is not None:
            assert d_best.is_cuda and d_best.dtype == torch.int32 and d_best.numel() >= 2
        if stream is None:
            stream = torch.cuda.current_stream(d_T.device).cuda_stream
        elif hasattr(stream, "cuda_stream"):
            stream = stream.cuda_stream
        _lib.check(self._lib.pgp_score_lcp_device(
            self._h, C.c_void_p(d_T.data_ptr()), n_h, int(mode), C.c_float(gate_deg),
            C.c_void_p(d_scores.data_ptr()),
            C.c_void_p(d_counts.data_ptr()) if d_counts is not None else None,
            C.c_void_p(d_best.data_ptr()) if d_best is not None else None,
            C.c_void_p(stream)))

    def settle_best_device(self, d_T, d_scores, d_best, mode=PGP_MODE_WEIGHTED, gate_deg=30.0, stream=None):
        """Arg-max (+ exact weighted near-tie settlement) over a complete device score vector that was
        assembled from several calls / several ranks; d_T holds ALL its transforms."""
        import torch
        n_h = int(d_T.shape[0])
        assert d_T.is_cuda and d_T.dtype == torch.float32 and d_T.is_contiguous()
        assert d_scores.is_cuda and d_scores.dtype == torch.float32 and d_scores.is_contiguous() and d_scores.numel() >= n_h
        assert d_best.is_cuda and d_best.dtype == torch.int32 and d_best.numel() >= 2
        if stream is None:
            stream = torch.cuda.current_stream(d_T.device).cuda_stream
        elif hasattr(stream, "cuda_stream"):
            stream = stream.cuda_stream
        _lib.check(self._lib.pgp_settle_best_device(
            self._h, C.c_void_p(d_T.data_ptr()), n_h, int(mode), C.c_float(gate_deg),
            C.c_void_p(d_scores.data_ptr()), C.c_void_p(d_best.data_ptr()), C.c_void_p(stream)))

    def set_kernel_timing(self, enable=True):
        """True / 1: time every scoring launch; N > 1: every Nth; False / 0: off."""
        _lib.check(self._lib.pgp_set_kernel_timing(self._h, int(enable)))

    def kernel_timing(self, reset=True):
        """(launches, total_ms) of the dominant kernel since the last reset (HIP events)."""
        n, ms = C.c_int(0), C.c_float(0)
        _lib.check(self._lib.pgp_get_kernel_timing(self._h, C.byref(n), C.byref(ms), int(reset)))
        return n.value, float(ms.value)

    def index_info(self):
        info = _lib.IndexInfo()
        _lib.check(self._lib.pgp_get_index_info(self._h, C.byref(info)))
        return {k: getattr(info, k) for k, _ in info._fields_}


class MultiGpuScorer:
    """One (scene, model) pair replicated on several GPUs of the node, hypotheses block-partitioned,
    scores combined with one RCCL all-reduce inside libpgp.so (pgp_multi_*, csrc/multi_gpu.hip):
    the single-process form of the sharding the node's callers see (SceneCfg.cpp:376-406)."""

    def __init__(self, device_ids=None):
        self._lib = _lib.load()
        h = C.c_void_p()
        if device_ids is None:
            ids, n = None, 0
        else:
            arr = np.ascontiguousarray(device_ids, np.int32)
            ids, n = arr.ctypes.data_as(_i), len(arr)
        _lib.check(self._lib.pgp_multi_create(C.byref(h), ids, n))
        self._h = h
        self.n_devices = int(self._lib.pgp_multi_size(h))
        self.nQ = 0
        self._n_up = 0
        self._n_up_obj = np.zeros(1, np.int32)

    @staticmethod
    def unique_id():
        """128 bytes from ncclGetUniqueId: one process of a ranked group draws it, all of them pass it to ranked()."""
        buf = C.create_string_buffer(128)
        _lib.check(_lib.load().pgp_multi_unique_id(buf))
        return buf.raw

    @classmethod
    def ranked(cls, device_ids, rank0, world, unique_id):
        """This process's members of a group that spans several processes (pgp_multi_create_ranked): the devices
        `device_ids` are ranks rank0 .. rank0 + len - 1 of `world`.  Blocks until every process has called it."""
        self = cls.__new__(cls)
        self._lib = _lib.load()
        arr = np.ascontiguousarray(device_ids, np.int32)
        h = C.c_void_p()
        _lib.check(self._lib.pgp_multi_create_ranked(C.byref(h), arr.ctypes.data_as(_i), len(arr), int(rank0), int(world),
                                                     C.c_char_p(bytes(unique_id))))
        self._h = h
        self.n_devices = int(self._lib.pgp_multi_size(h))
        self.nQ = 0
        self._n_up = 0
        self._n_up_obj = np.zeros(1, np.int32)
        self._slot_n = {}
        return self

    def member(self, k=0):
        """Member k's context of object 0 as a (borrowed) LcpScorer: kernel timing, index info."""
        return LcpScorer.borrowed(self._lib.pgp_multi_context(self._h, int(k)))

    def info(self):
        inf = _lib.MultiInfo()
        _lib.check(self._lib.pgp_multi_get_info(self._h, C.byref(inf)))
        return {"n_local": inf.n_local, "world": inf.world, "rank0": inf.rank0, "rccl_ranks": inf.rccl_ranks,
                "emulated": bool(inf.emulated), "devices": list(inf.devices[:min(inf.n_local, 16)]), "exchanges": int(inf.exchanges)}

    # ---- streaming form: resident batches, steps queued without a host wait (pgp_multi_enqueue_slot) ----
    def upload_slot(self, slot, T):
        T = _f32(T, 16)
        _lib.check(self._lib.pgp_multi_upload_slot(self._h, int(slot), _fp(T), len(T)))
        if not hasattr(self, "_slot_n"):
            self._slot_n = {}
        self._slot_n[int(slot)] = len(T)

    def enqueue_slot(self, slot, mode=PGP_MODE_PLAIN, gate_deg=30.0):
        _lib.check(self._lib.pgp_multi_enqueue_slot(self._h, int(slot), int(mode), C.c_float(gate_deg)))
        self._last_slot = int(slot)

    def collect(self):
        """Completes every queued step; (scores, counts, best_index, best_score) of the LAST one."""
        s, c, bi, bs = self._out(self._slot_n[self._last_slot])
        _lib.check(self._lib.pgp_multi_collect(self._h, _fp(s), c.ctypes.data_as(_i), C.byref(bi), C.byref(bs)))
        return s, c, bi.value, float(np.float32(bs.value))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pgp_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def slice_of(n_total, k, n_dev):
        lo, hi = C.c_int(0), C.c_int(0)
        _lib.check(_lib.load().pgp_multi_slice(int(n_total), int(k), int(n_dev), C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def init(self, P_xyz, P_nrm, P_w, Q_xyz, Q_nrm, delta=0.005):
        xyz, nrm, w = _f32(P_xyz, 3), _f32(P_nrm, 3), _f32(P_w)
        _lib.check(self._lib.pgp_multi_set_scene(self._h, _fp(xyz), _fp(nrm), _fp(w), len(xyz), C.c_float(delta)))
        q, qn = _f32(Q_xyz, 3), _f32(Q_nrm, 3)
        _lib.check(self._lib.pgp_multi_set_model(self._h, _fp(q), _fp(qn), len(q)))
        self.nQ = len(q)

    def _out(self, n_h):
        return np.zeros(n_h, np.float32), np.zeros(n_h, np.int32), C.c_int(-1), C.c_float(0)

    def score(self, T, mode=PGP_MODE_PLAIN, gate_deg=30.0):
        """Same contract as LcpScorer.score: (scores, counts, best_index, best_score)."""
        T = _f32(T, 16)
        s, c, bi, bs = self._out(len(T))
        _lib.check(self._lib.pgp_multi_score_lcp(self._h, _fp(T), len(T), int(mode), C.c_float(gate_deg), _fp(s),
                                                 c.ctypes.data_as(_i), C.byref(bi), C.byref(bs)))
        return s, c, bi.value, float(np.float32(bs.value))

    def upload(self, T):
        T = _f32(T, 16)
        _lib.check(self._lib.pgp_multi_upload(self._h, _fp(T), len(T)))
        self._n_up = len(T)
        self._n_up_obj = np.array([len(T)], np.int32)

    def score_uploaded(self, mode=PGP_MODE_PLAIN, gate_deg=30.0):
        s, c, bi, bs = self._out(self._n_up)
        _lib.check(self._lib.pgp_multi_score_uploaded(self._h, int(mode), C.c_float(gate_deg), _fp(s),
                                                      c.ctypes.data_as(_i), C.byref(bi), C.byref(bs)))
        return s, c, bi.value, float(np.float32(bs.value))

    def set_exact_records(self, on=True):
        """pgp_set_exact_records on member 0's context: the group's running-best list is then the reference's."""
        ctx0 = self._lib.pgp_multi_context(self._h, 0)
        _lib.check(self._lib.pgp_set_exact_records(C.c_void_p(ctx0), 1 if on else 0))

    def set_verify_early_out(self, on=True):
        ctx0 = self._lib.pgp_multi_context(self._h, 0)
        _lib.check(self._lib.pgp_set_verify_early_out(C.c_void_p(ctx0), 1 if on else 0))

    def last_timing(self):
        a, b, c = C.c_float(0), C.c_float(0), C.c_float(0)
        _lib.check(self._lib.pgp_multi_last_timing(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"upload_ms": a.value, "enqueue_ms": b.value, "total_ms": c.value}

    # ---- several objects in one group (configs[3]; SceneCfg.cpp:376-406) ------------------------------
    @property
    def n_objects(self):
        return int(self._lib.pgp_multi_objects(self._h))

    def add_object(self):
        """pgp_multi_add_object: a further (scene, model) pair replicated on every member; returns its id."""
        o = int(self._lib.pgp_multi_add_object(self._h))
        if o < 0:
            _lib.check(o)
        return o

    def object_context(self, obj, k=0):
        return C.c_void_p(self._lib.pgp_multi_object_context(self._h, int(obj), int(k)))

    def init_object(self, obj, P_xyz, P_nrm, P_w, Q_xyz, Q_nrm, delta=0.005):
        xyz, nrm, w = _f32(P_xyz, 3), _f32(P_nrm, 3), _f32(P_w)
        _lib.check(self._lib.pgp_multi_set_object_scene(self._h, int(obj), _fp(xyz), _fp(nrm), _fp(w), len(xyz), C.c_float(delta)))
        q, qn = _f32(Q_xyz, 3), _f32(Q_nrm, 3)
        _lib.check(self._lib.pgp_multi_set_object_model(self._h, int(obj), _fp(q), _fp(qn), len(q)))

    def set_object_search_model(self, obj, xyz):
        q = _f32(xyz, 3)
        _lib.check(self._lib.pgp_multi_set_object_search_model(self._h, int(obj), _fp(q), len(q)))

    def set_object_ppf_map(self, obj, keys, counts=None, pairs=None):
        k = np.ascontiguousarray(keys, np.int32).reshape(-1, 4)
        c = None if counts is None else np.ascontiguousarray(counts, np.int32)
        p = None if pairs is None else np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
        _lib.check(self._lib.pgp_multi_set_object_ppf_map(
            self._h, int(obj), k.ctypes.data_as(_i), None if c is None else c.ctypes.data_as(_i),
            None if p is None else p.ctypes.data_as(_i), len(k)))

    @staticmethod
    def flat_slices(counts, k, n_dev):
        """pgp_multi_flat_slices: member k's share of the flat (object, unit) space as [(object, lo, hi), ...]."""
        c = np.ascontiguousarray(counts, np.int32)
        n = len(c)
        o, lo, hi = (np.zeros(max(n, 1), np.int32) for _ in range(3))
        m = C.c_int(0)
        _lib.check(_lib.load().pgp_multi_flat_slices(c.ctypes.data_as(_i), n, int(k), int(n_dev), o.ctypes.data_as(_i),
                                                     lo.ctypes.data_as(_i), hi.ctypes.data_as(_i), C.byref(m)))
        return [(int(o[i]), int(lo[i]), int(hi[i])) for i in range(m.value)]

    def _lists(self, T_per_object):
        Ts = [_f32(T, 16) for T in T_per_object]
        n = np.array([len(T) for T in Ts], np.int32)
        ptrs = (_f * len(Ts))(*[_fp(T) for T in Ts])
        return Ts, n, ptrs

    def _flat_out(self, n, n_obj):
        N = int(n.sum())
        return (np.zeros(N, np.float32), np.zeros(N, np.int32), np.full(n_obj, -1, np.int32), np.zeros(n_obj, np.float32))

    def _split(self, n, s, c, bi, bs):
        off = np.concatenate([[0], np.cumsum(n)])
        return [(s[off[o]:off[o + 1]], c[off[o]:off[o + 1]], int(bi[o]), float(bs[o])) for o in range(len(n))]

    def score_objects(self, T_per_object, mode=PGP_MODE_PLAIN, gate_deg=30.0):
        """pgp_multi_score_objects: per object (scores, counts, best_index, best_score), as LcpScorer.score returns them."""
        Ts, n, ptrs = self._lists(T_per_object)
        s, c, bi, bs = self._flat_out(n, len(Ts))
        _lib.check(self._lib.pgp_multi_score_objects(self._h, ptrs, n.ctypes.data_as(_i), len(Ts), int(mode), C.c_float(gate_deg),
                                                     _fp(s), c.ctypes.data_as(_i), bi.ctypes.data_as(_i), _fp(bs)))
        return self._split(n, s, c, bi, bs)

    def upload_objects(self, T_per_object):
        Ts, n, ptrs = self._lists(T_per_object)
        _lib.check(self._lib.pgp_multi_upload_objects(self._h, ptrs, n.ctypes.data_as(_i), len(Ts)))
        self._n_up_obj = n
        self._n_up = int(n.sum())

    def score_objects_uploaded(self, mode=PGP_MODE_PLAIN, gate_deg=30.0):
        n = self._n_up_obj
        s, c, bi, bs = self._flat_out(n, len(n))
        _lib.check(self._lib.pgp_multi_score_objects_uploaded(self._h, int(mode), C.c_float(gate_deg), _fp(s), c.ctypes.data_as(_i),
                                                              bi.ctypes.data_as(_i), _fp(bs)))
        return self._split(n, s, c, bi, bs)

    # ---- ICP pose shards (UCTSearch.cpp:200-266 -> UCTState.cpp:121-204) ------------------------------
    def icp_refine(self, jobs, trim=1.0, max_iterations=100, max_corr_dist=0.0, energy_ratio=1.0):
        """pgp_multi_icp_refine.  jobs: [(src_xyz, tgt_xyz, T), ...]; returns per job (T, energy, iters) as
        LcpScorer.icp_refine does."""
        arr = (_lib.MultiIcpJob * max(len(jobs), 1))()
        keep, out = [], []
        for j, (src, tgt, T) in enumerate(jobs):
            src, tgt = _f32(src, 3), _f32(tgt, 3)
            T = _f32(T, 16).copy()
            e, it = np.zeros(len(T), np.float32), np.zeros(len(T), np.int32)
            keep.append((src, tgt))
            out.append((T, e, it))
            arr[j] = _lib.MultiIcpJob(_fp(src), len(src), _fp(tgt), len(tgt), _fp(T), len(T), _fp(e), it.ctypes.data_as(_i))
        prm = _lib.IcpParams(int(max_iterations), float(trim), float(max_corr_dist), float(energy_ratio))
        _lib.check(self._lib.pgp_multi_icp_refine(self._h, arr, len(jobs), C.byref(prm)))
        return out

    # ---- congruent sets sharded by base (base.cc:1855-1874) -------------------------------------------
    def find_congruent_batch(self, obj, base_ids, base_xyz, invariants, threshold):
        ids = np.ascontiguousarray(base_ids, np.int32).reshape(-1, 4)
        xyz = _f32(np.asarray(base_xyz).reshape(-1, 12))
        inv = _f32(invariants, 2)
        nq = np.zeros(max(len(ids), 1), np.int32)
        _lib.check(self._lib.pgp_multi_find_congruent_batch(self._h, int(obj), ids.ctypes.data_as(_i), _fp(xyz), _fp(inv), len(ids),
                                                            C.c_float(threshold), nq.ctypes.data_as(_i)))
        return nq[:len(ids)]

    def congruent_batch_quads(self, obj, picks):
        p = np.ascontiguousarray(picks, np.int32).reshape(-1, 2)
        q = np.zeros((max(len(p), 1), 4), np.int32)
        _lib.check(self._lib.pgp_multi_congruent_batch_quads(self._h, int(obj), p.ctypes.data_as(_i), len(p), q.ctypes.data_as(_i)))
        return q[:len(p)]

    def congruent_batch_fit(self, obj, picks, base_ids, centroid_P, centroid_Q):
        p = np.ascontiguousarray(picks, np.int32).reshape(-1, 2)
        ids = np.ascontiguousarray(base_ids, np.int32).reshape(-1, 4)
        m = len(p)
        T = np.zeros((max(m, 1), 16), np.float32)
        pose = np.zeros((max(m, 1), 16), np.float64)
        st = np.zeros(max(m, 1), np.int32)
        rms = np.zeros(max(m, 1), np.float32)
        cP, cQ = _f32(centroid_P), _f32(centroid_Q)
        _lib.check(self._lib.pgp_multi_congruent_batch_fit(self._h, int(obj), p.ctypes.data_as(_i), ids.ctypes.data_as(_i), m, _fp(cP),
                                                           _fp(cQ), _fp(T), pose.ctypes.data_as(C.POINTER(C.c_double)),
                                                           st.ctypes.data_as(_i), _fp(rms)))
        return T[:m], pose[:m], st[:m], rms[:m]
