#!/usr/bin/env python3
"""Randomised parity soak for the rows beside the scoring loop: congruent sets (pairs as sets, quads
in order), rigid fits (bit-exact centred transform / status / rms), pose clustering (identical
representatives and assignments), back-projection (bit-exact list) -- libpgp.so against the CPU
oracle for FUZZ_SECONDS (default 90).  Test infrastructure: lives under tests/ because it uses the oracle as its checker; run by hand
(python tests/soak_parity.py), not collected by pytest."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.dirname(os.path.abspath(__file__))]
from physimglobalpose_amd import LcpScorer, synth  # noqa: E402
from _checkers import (CongruentChecker, oracle_backproject, oracle_greedy_cluster,  # noqa: E402
                       oracle_rigid_from_pairs)


def main():
    budget = float(os.environ.get("FUZZ_SECONDS", "90"))
    t0, n = time.time(), 0
    sc = LcpScorer(0)
    while time.time() - t0 < budget:
        rng = np.random.default_rng(5000 + n)
        # ---- congruent sets
        ns = int(rng.choice([60, 150, 400, 700]))
        w = synth.make_workload(3000, 800, 4, config_id=2000 + n, n_search=ns)
        sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
        sc.set_search_model(w.Qs_xyz)
        orc = CongruentChecker(w.Qs_xyz, "oracle")
        T = w.T_gt.reshape(4, 4).T
        ids = rng.choice(ns, 4, replace=False)
        base = (w.Qs_xyz[ids] @ T[:3, :3].T + T[:3, 3] + 0.0005 * rng.standard_normal((4, 3))).astype(np.float32)
        d1, d6 = (float(np.linalg.norm(base[a] - base[b])) for a, b in ((0, 1), (2, 3)))
        eps = float(w.delta * rng.choice([0.5, 1.0, 2.0]))
        p1, p6 = sc.extract_pairs(d1, eps, cap=1 << 21), sc.extract_pairs(d6, eps, cap=1 << 21)
        o1, o6 = orc.extract_pairs(d1, eps), orc.extract_pairs(d6, eps)
        assert set(map(tuple, p1.tolist())) == set(map(tuple, o1.tolist())), f"pairs, case {n}"
        assert set(map(tuple, p6.tolist())) == set(map(tuple, o6.tolist())), f"pairs, case {n}"
        i1, i2 = float(rng.uniform(0.1, 0.9)), float(rng.uniform(0.1, 0.9))
        q = sc.find_congruent(base, i1, i2, eps, o1, o6, cap=1 << 21)
        qo = orc.find_congruent(base, i1, i2, eps, o1, o6)
        assert np.array_equal(q, qo), f"quads, case {n}"
        # ---- rigid fits on random (base, quad) index pairs
        m = 500
        b = rng.integers(0, len(w.P_xyz), (m, 4)).astype(np.int32)
        qd = rng.integers(0, ns, (m, 4)).astype(np.int32)
        if len(q):
            qd[: min(m, len(q))] = q[: min(m, len(q))]
        Tg, pose, st, rms = sc.rigid_from_congruent(b, qd, w.centroid_P, w.centroid_Q)
        To, po, so, ro = oracle_rigid_from_pairs(w.P_xyz, w.Qs_xyz, b, qd, w.centroid_P, w.centroid_Q)
        assert np.array_equal(st, so), f"rigid status, case {n}"
        ok = st == 1
        assert np.array_equal(Tg[ok], To[ok]) and np.array_equal(rms[ok], ro[ok]), f"rigid bits, case {n}"
        assert np.allclose(pose[ok], po[ok], rtol=0, atol=1e-5), f"rigid pose, case {n}"
        # ---- clustering of random scored poses
        k = int(rng.choice([1, 40, 300, 1500]))
        Tc = np.stack([synth.colmajor16(synth._se3(synth._random_rot(rng, np.deg2rad(rng.choice([3, 20, 180]))),
                                                  rng.normal(0, rng.choice([0.005, 0.05]), 3))) for _ in range(k)])
        s = (rng.integers(0, 50, k) / 50.0).astype(np.float32)
        sym = rng.choice([0, 90, 180, 360], 3).astype(np.float32)
        rep, asg = sc.cluster_poses(Tc, s, float(s.max()), sym)
        rep_o, asg_o = oracle_greedy_cluster(Tc, s, float(s.max()), sym)
        assert np.array_equal(rep, rep_o) and np.array_equal(asg, asg_o), f"cluster, case {n}"
        # ---- back-projection of a random 16-bit image
        rows, cols = int(rng.integers(1, 300)), int(rng.integers(1, 400))
        raw = rng.integers(0, 65536, (rows, cols)).astype(np.uint16)
        mask = (rng.random((rows, cols)) < 0.5).astype(np.uint8)
        K = np.array([[rng.uniform(200, 900), 0, cols / 2], [0, rng.uniform(200, 900), rows / 2], [0, 0, 1]], np.float32)
        assert np.array_equal(sc.backproject_depth(raw, K, mask), oracle_backproject(raw, mask, K)), f"backproject, case {n}"
        n += 1
        if n % 100 == 0:   # a silent GPU job is taken for hung after a few minutes
            print(f"... {n} cases, {time.time() - t0:.0f} s", flush=True)
    print(f"fuzz ok: {n} random cases of congruent sets / rigid fits / clustering / back-projection, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
