#!/usr/bin/env python3
"""Randomised parity soak: libpgp.so against the CPU oracle over random scenes, models, radii,
hypothesis mixes and both modes, for FUZZ_SECONDS (default 120).  Exact equality for plain counts /
scores / best index, 2e-6 absolute for weighted scores.  Test infrastructure: lives under tests/ because it uses the oracle as its checker; run by hand
(python tests/soak_parity.py), not collected by pytest."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.dirname(os.path.abspath(__file__))]
from physimglobalpose_amd import LcpScorer, PGP_MODE_PLAIN, PGP_MODE_WEIGHTED, synth  # noqa: E402
from _checkers import Oracle, oracle_lib  # noqa: E402


def main():
    budget = float(os.environ.get("FUZZ_SECONDS", "120"))
    seed0 = int(os.environ.get("FUZZ_SEED", "1"))
    threads = int(oracle_lib().orc_max_threads())
    t0, n_cases, n_hyp = time.time(), 0, 0
    sc = LcpScorer(0)
    while time.time() - t0 < budget:
        seed = seed0 + n_cases
        rng = np.random.default_rng(seed)
        n_scene = int(rng.choice([50, 300, 2000, 8000, 20000]))
        n_model = int(rng.choice([1, 7, 63, 64, 65, 255, 257, 900, 3000]))
        n_h = int(rng.choice([1, 3, 17, 64, 200]))
        w = synth.make_workload(max(n_scene, 40), max(n_model, 8), n_h, config_id=1000 + seed)
        Q, Qn = w.Q_xyz[:n_model], w.Q_nrm[:n_model]
        delta = float(w.delta * rng.choice([0.2, 0.5, 1.0, 2.0, 6.0]))
        T = w.T.copy()
        if rng.random() < 0.3:      # far / degenerate transforms mixed in
            T[rng.integers(n_h)] = synth.colmajor16(synth._se3(synth._random_rot(rng), rng.uniform(-50, 50, 3)))
        if rng.random() < 0.2:
            T[rng.integers(n_h)][12] = np.nan
        P, Pn, Pw = w.P_xyz, w.P_nrm, w.P_w
        if rng.random() < 0.3:      # duplicated scene points: exact distance ties
            k = len(P) // 10
            P, Pn, Pw = np.concatenate([P, P[:k]]), np.concatenate([Pn, Pn[:k]]), np.concatenate([Pw, Pw[:k]])
        sc.init(P, Pn, Pw, Q, Qn, delta)
        orc = Oracle(P, Pn, Pw, Q, Qn)
        s, c, bi, _ = sc.score(T, PGP_MODE_PLAIN)
        so, bio, _ = orc.score_batch(T, delta, mode=0, threads=threads)
        assert np.array_equal(s, so) and bi == bio, f"plain mismatch, seed {seed}"
        ties = len(P) != len(w.P_xyz)
        if not ties:                # with duplicates the NN id (hence normal / weight) may differ
            gate = float(rng.choice([10.0, 30.0, 60.0, 90.0]))
            sw, _, biw, bsw = sc.score(T, PGP_MODE_WEIGHTED, gate)
            swo, biwo, _ = orc.score_batch(T, delta, mode=1, gate_deg=gate, threads=threads)
            assert np.allclose(sw, swo, rtol=0, atol=2e-6), f"weighted mismatch, seed {seed}"
            # the returned best pose is the reference's, near-ties included (exact settlement on the device)
            assert biw == biwo and (biw < 0 or abs(bsw - swo[biw]) <= 2e-6), f"weighted best mismatch, seed {seed}"
        n_cases += 1
        if n_cases % 100 == 0:   # a silent GPU job is taken for hung after a few minutes
            print(f"... {n_cases} cases, {time.time() - t0:.0f} s", flush=True)
        n_hyp += n_h
    print(f"fuzz ok: {n_cases} random cases, {n_hyp} hypotheses, {time.time() - t0:.0f} s, seeds {seed0}..{seed0 + n_cases - 1}")


if __name__ == "__main__":
    main()
