"""The C restatement of pcl::MovingLeastSquares (oracle/pgp_oracle.c orc_mls) against an independent
numpy implementation of the same published algorithm (tests/golden/make_mls_golden.py -> mls.npz).
PCL itself is absent from this image: two codes that share neither lines nor numerical methods
agreeing to float resolution is the pin there is."""
import os

import numpy as np

from _checkers import oracle_mls

GOLD = os.path.join(os.path.dirname(__file__), "golden", "mls.npz")


def check_against_fixture(got, g, pos_tol=3e-7, nrm_tol=2e-5):
    ox, on, oc, oi = got
    assert np.array_equal(oi, g["out_index"])                     # same points dropped (< 3 neighbours)
    assert np.abs(ox.astype(np.float64) - g["out_xyz"]).max() <= pos_tol
    # the plane normal's sign is the eigen-solver's choice: compare directions, not signs
    sgn = np.sign(np.einsum("ij,ij->i", on.astype(np.float64), g["out_nrm"]))
    assert np.all(sgn != 0)
    assert np.abs(on.astype(np.float64) * sgn[:, None] - g["out_nrm"]).max() <= nrm_tol
    assert np.allclose(oc, g["out_curv"], rtol=2e-4, atol=1e-9)


def test_c_restatement_agrees_with_the_independent_implementation():
    g = np.load(GOLD)
    assert (g["n_neighbours"] < 6).sum() > 0 and (g["n_neighbours"] >= 6).sum() > 300   # both branches present
    check_against_fixture(oracle_mls(g["xyz"], float(g["radius"])), g)


def test_small_and_empty_clouds():
    assert len(oracle_mls(np.zeros((0, 3), np.float32))[3]) == 0
    two = np.array([[0, 0, 0.5], [0.005, 0, 0.5]], np.float32)
    assert len(oracle_mls(two)[3]) == 0                            # fewer than 3 neighbours: dropped
    tri = np.array([[0, 0, 0.5], [0.005, 0, 0.5], [0, 0.005, 0.5]], np.float32)
    ox, on, oc, oi = oracle_mls(tri)
    assert list(oi) == [0, 1, 2]
    assert np.allclose(np.abs(on), [[0, 0, 1]] * 3, atol=1e-6)     # three points: the plane through them
    assert np.allclose(ox, tri, atol=1e-7) and np.allclose(oc, 0, atol=1e-6)
