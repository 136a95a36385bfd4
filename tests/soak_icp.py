#!/usr/bin/env python3
"""Randomised soak of the ICP paths: the default path (LDS index + vicinity graph, clustered / plain / multi-target
launches) against the exhaustive scan (PGP_ICP_NN=scan), transforms, energies and iteration counts bit for bit, for
FUZZ_SECONDS (default 120).  The vicinity graph's answer rests on float margins (csrc/icp.hip nnidx_vic_*): the soak
throws at it regular lattices (exact ties everywhere), duplicated target points, thin sheets a millimetre apart,
clouds scaled from millimetres to tens of metres, far outliers, sources that coincide with targets.
Test infrastructure (run by hand: python tests/soak_icp.py), not collected by pytest."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.dirname(os.path.abspath(__file__))]
from physimglobalpose_amd import LcpScorer, synth  # noqa: E402


def target(rng, kind, n):
    if kind == 0:
        return synth.make_model(rng, n)[0]
    if kind == 1:      # lattice: equal distances everywhere
        k = int(round(n ** (1 / 3))) + 1
        g = np.stack(np.meshgrid(*[np.arange(k)] * 3, indexing="ij"), -1).reshape(-1, 3)[:n]
        return g * 0.01
    if kind == 2:      # duplicates
        M = rng.uniform(-0.1, 0.1, (n, 3))
        M[n // 2:] = M[:n - n // 2]
        return M
    if kind == 3:      # two thin sheets 1 mm apart
        M = np.c_[rng.uniform(-0.1, 0.1, (n, 2)), np.where(rng.random(n) < 0.5, 0.0, 0.001)]
        return M
    M = rng.uniform(-0.1, 0.1, (n, 3))      # uniform volume
    return M


def main():
    budget = float(os.environ.get("FUZZ_SECONDS", "120"))
    t0, n_case, n_pose_it = time.time(), 0, 0
    sc = LcpScorer(0)
    while time.time() - t0 < budget:
        rng = np.random.default_rng(9000 + n_case)
        n_tgt = int(rng.choice([9, 70, 500, 2000, 5000, 6000]))
        n_src = int(rng.choice([5, 100, 900, 1800, 2500, 4000]))
        n_pose = int(rng.choice([1, 5, 40, 64, 100, 200, 300]))
        kind = int(rng.integers(0, 5))
        scale = float(rng.choice([0.01, 1.0, 1.0, 30.0]))
        M = (target(rng, kind, n_tgt) * scale).astype(np.float32)
        R = synth._random_rot(rng)
        t = rng.uniform(-0.5, 0.5, 3) * scale
        noise = float(rng.choice([0.0, 0.0003, 0.002])) * scale
        S = M[rng.integers(0, len(M), n_src)].astype(np.float64) @ R.T + t + noise * rng.standard_normal((n_src, 3))
        if n_case % 3 == 0:
            S[rng.integers(0, n_src, max(1, n_src // 12))] += rng.uniform(-0.2, 0.2, 3) * scale
        S = S.astype(np.float32)
        Tinv = np.linalg.inv(synth._se3(R, t))
        rot = float(rng.choice([0.0, 0.3, 4.0, 15.0]))
        G = np.stack([synth.colmajor16(Tinv @ synth._se3(synth._random_rot(rng, np.deg2rad(rot)), 0.002 * rot * scale * rng.standard_normal(3)))
                      for _ in range(n_pose)])
        form = [dict(max_iterations=14, trim_fraction=0.9, energy_ratio=1.0),
                dict(max_iterations=9, trim_fraction=0.55, energy_ratio=0.0),
                dict(max_iterations=10, max_corr_dist=0.03 * scale, energy_ratio=0.0, transformation_epsilon=1e-9, absolute_mse=1e-14),
                dict(max_iterations=10, trim_fraction=1.0, energy_ratio=1.0)][n_case % 4]
        os.environ["PGP_ICP_NN"] = "scan"
        ref = sc.icp_refine_ex(S, M, G, **form)
        del os.environ["PGP_ICP_NN"]
        got = sc.icp_refine_ex(S, M, G, **form)
        for x, y, what in zip(ref, got, ("T", "energy", "iters")):
            if not np.array_equal(x, y, equal_nan=True):
                print(f"MISMATCH case {n_case}: {what}; n_tgt {len(M)} n_src {n_src} poses {n_pose} kind {kind} scale {scale} rot {rot} form {form}")
                sys.exit(1)
        n_case += 1
        n_pose_it += int(np.asarray(ref[2]).sum())
    print(f"{n_case} random ICP problems, {n_pose_it} pose-iterations: default path == exhaustive scan, bit for bit ({time.time() - t0:.0f} s)")


if __name__ == "__main__":
    main()
