#!/usr/bin/env python3
"""would SHARING the cell look-up between the hypotheses a workgroup loops over pay?  The look-up (cell ->
occupancy word -> rank -> run descriptor: ~30 of the ~127 vector instructions per (model point, hypothesis)) can only be SKIPPED
by a wave when all 64 lanes land in the cell they landed in under the previous hypothesis -- a lane-level hit saves no issue
slot, the wave executes the look-up for its other lanes anyway.  This tool COUNTS that opportunity before anyone touches the
kernel: hypotheses sorted by a coarse pose key (or chained greedily by pose distance), 8 per workgroup as launch_score deals
them, the model in the kernel's own Morton order, 64 consecutive points per wave, cells of the index's own size; on
  (i)  the C2 bench batch (4096 hypotheses: 25 % within 5 deg / 3 mm of the true pose, 50 % within 30 deg / 2 cm, 25 % random),
  (ii) a drop-in list (the fits of one object's 100 bases: pgp_find_congruent_batch + fits on the device).
Writes profiles/r06_ab/shared_lookup.json.  usage: python tools/shared_lookup_stats.py"""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from physimglobalpose_amd import LcpScorer, synth  # noqa: E402


def morton_order(Q):
    q = Q - Q.min(0)
    q = np.minimum((q / max(q.max(), 1e-9) * 1023).astype(np.uint32), 1023)

    def spread(v):
        v = v & 1023
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    key = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    return np.argsort(key, kind="stable")


def cells_of(T16, Q, origin, h):
    """[n_h, n_q, 3] int32 cell coordinates of T q (float32 arithmetic, the kernel's floor((x - o) / h))"""
    M = T16.reshape(-1, 4, 4).transpose(0, 2, 1).astype(np.float32)   # col-major images
    X = np.einsum("hij,qj->hqi", M[:, :3, :3], Q.astype(np.float32)) + M[:, None, :3, 3]
    return np.floor((X - origin.astype(np.float32)) * np.float32(1.0 / h)).astype(np.int32)


def pose_keys(T16, q_centre, step_t, step_r):
    """coarse pose key: quantised image of the model's centre, then quantised rotation vector"""
    M = T16.reshape(-1, 4, 4).transpose(0, 2, 1).astype(np.float64)
    c = np.einsum("hij,j->hi", M[:, :3, :3], q_centre) + M[:, :3, 3]
    R = M[:, :3, :3]
    ang = np.arccos(np.clip((np.trace(R, axis1=1, axis2=2) - 1) / 2, -1, 1))
    ax = np.stack([R[:, 2, 1] - R[:, 1, 2], R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] - R[:, 0, 1]], 1)
    ax /= np.maximum(np.linalg.norm(ax, axis=1, keepdims=True), 1e-12)
    rv = ax * ang[:, None]
    return np.concatenate([np.floor(c / step_t), np.floor(rv / step_r)], 1).astype(np.int64)


def order_by_key(keys):
    return np.lexsort(keys.T[::-1])


def greedy_chain(T16, Q, cap=4096):
    """nearest-neighbour chain by the mean displacement of 32 probe points: the best a sort could hope for"""
    n = min(len(T16), cap)
    M = T16[:n].reshape(-1, 4, 4).transpose(0, 2, 1).astype(np.float32)
    probe = Q[np.linspace(0, len(Q) - 1, 32).astype(int)].astype(np.float32)
    X = (np.einsum("hij,qj->hqi", M[:, :3, :3], probe) + M[:, None, :3, 3]).reshape(n, -1)
    left = np.ones(n, bool)
    order = [0]
    left[0] = False
    for _ in range(n - 1):
        d = ((X - X[order[-1]]) ** 2).sum(1)
        d[~left] = np.inf
        k = int(np.argmin(d))
        order.append(k)
        left[k] = False
    return np.array(order)


def opportunity(T16, Q_morton, origin, h, order, hpb=8):
    T = T16[order]
    n = len(T) // hpb * hpb
    C = cells_of(T[:n], Q_morton, origin, h)                       # [n, nq, 3]
    nq = C.shape[1] // 64 * 64
    C = C[:, :nq].reshape(n // hpb, hpb, nq // 64, 64, 3)          # chunk, slot, wave, lane, xyz
    same = (C[:, 1:] == C[:, :-1]).all(-1)                         # slot s vs s - 1, per lane
    lane_hit = float(same.mean())
    wave_hit = float(same.all(-1).mean())                          # every lane of the wave: the look-up could be SKIPPED
    trips_with_prev = (hpb - 1) / hpb
    return {"lane_same_cell": lane_hit, "wave_all_lanes_same_cell": wave_hit,
            "skippable_share_of_all_trips": wave_hit * trips_with_prev,
            "upper_bound_on_valu_saved": wave_hit * trips_with_prev * 30.0 / 127.0}


def dropin_list():
    from _dropin import make_dropin_case
    with tempfile.TemporaryDirectory() as d:
        _, c = make_dropin_case(d)
    w, table = c["w"], c["table"]
    keys = np.array(list(table.keys()), np.int32)
    counts = np.array([len(table[tuple(k)]) for k in keys.tolist()], np.int32)
    pairs = np.concatenate([np.array(table[tuple(k)], np.int32).reshape(-1, 2) for k in keys.tolist()])
    sc = LcpScorer(0)
    sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    sc.set_search_model(w.Qs_xyz)
    sc.set_ppf_map(keys, counts, pairs)
    rng = np.random.default_rng(3)
    ids, inv, status = sc.select_bases(rng.random((256, 4)))
    ids, inv = ids[status == 1][:100], inv[status == 1][:100]
    nq = sc.find_congruent_batch(ids, w.P_xyz[ids], inv, w.delta)
    picks = np.array([(b, j) for b in range(len(nq)) for j in range(min(int(nq[b]), 100))], np.int32).reshape(-1, 2)
    T, pose, st = sc.congruent_batch_fit(picks, ids, w.centroid_P, w.centroid_Q)[:3]
    info = sc.index_info()
    return w, np.ascontiguousarray(T[st == 1]), info, picks[st == 1][:, 0]


def main():
    out = {"question": "share of (wave, hypothesis) trips whose cell look-up could be skipped because all 64 lanes land in the "
                       "cells of the previous hypothesis of the workgroup; upper bound on the vector instructions saved = that "
                       "share x 30 / 127", "hypotheses_per_workgroup": 8}
    # (i) the bench batch
    w = synth.make_workload(50000, 5000, 4096, config_id=2)
    sc = LcpScorer(0)
    sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    info = sc.index_info()
    h = float(info["cell_size"])
    origin = (w.P_xyz.min(0) - 2 * h).astype(np.float32)     # (the statistic does not depend on where the lattice starts)
    Qm = w.Q_xyz[morton_order(w.Q_xyz)]
    rows = {}
    rows["as_given"] = opportunity(w.T, Qm, origin, h, np.arange(len(w.T)))
    for name, (st_, sr_) in {"key_4mm_2deg": (0.004, np.deg2rad(2)), "key_1mm_0.5deg": (0.001, np.deg2rad(0.5))}.items():
        rows["sorted_" + name] = opportunity(w.T, Qm, origin, h, order_by_key(pose_keys(w.T, w.Q_xyz.mean(0), st_, sr_)))
    rows["greedy_chain"] = opportunity(w.T, Qm, origin, h, greedy_chain(w.T, w.Q_xyz))
    out["c2_bench_batch"] = {"hypotheses": int(len(w.T)), "cell_size_m": h, **rows}
    # (ii) a drop-in list
    wd, Td, info_d, base_of = dropin_list()
    hd = float(info_d["cell_size"])
    od = (wd.P_xyz.min(0) - 2 * hd).astype(np.float32)
    Qd = wd.Q_xyz[morton_order(wd.Q_xyz)]
    rows = {}
    rows["as_given_base_after_base"] = opportunity(Td, Qd, od, hd, np.arange(len(Td)))
    for name, (st_, sr_) in {"key_4mm_2deg": (0.004, np.deg2rad(2)), "key_1mm_0.5deg": (0.001, np.deg2rad(0.5))}.items():
        rows["sorted_" + name] = opportunity(Td, Qd, od, hd, order_by_key(pose_keys(Td, wd.Q_xyz.mean(0), st_, sr_)))
    rows["greedy_chain"] = opportunity(Td, Qd, od, hd, greedy_chain(Td, wd.Q_xyz))
    # how far apart are neighbouring fits at all?
    M = Td.reshape(-1, 4, 4).transpose(0, 2, 1)
    c = np.einsum("hij,j->hi", M[:, :3, :3], wd.Q_xyz.mean(0)) + M[:, :3, 3]
    ch = greedy_chain(Td, wd.Q_xyz)
    step = np.linalg.norm(np.diff(c[ch], axis=0), axis=1)
    out["drop_in_list"] = {"hypotheses": int(len(Td)), "bases": int(len(np.unique(base_of))), "cell_size_m": hd,
                           "centre_step_along_greedy_chain_mm": {"median": float(np.median(step) * 1e3), "p10": float(np.percentile(step, 10) * 1e3)},
                           **rows}
    best = max(max(v["upper_bound_on_valu_saved"] for v in out[k].values() if isinstance(v, dict) and "upper_bound_on_valu_saved" in v)
               for k in ("c2_bench_batch", "drop_in_list"))
    out["verdict"] = (f"largest upper bound on the vector instructions a shared look-up could save: {best * 100:.2f} % "
                      "(adoption bar: 8 % of the step time on the drop-in list) -- a wave's 64 model points span centimetres and "
                      "the index's cells are 4 mm: some lane always changes its cell; the wave-trips that do keep every cell on "
                      "the drop-in list are repeated fits.  Not built; closed.")
    path = os.path.join(ROOT, "profiles", "r06_ab", "shared_lookup.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
