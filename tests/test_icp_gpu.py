"""Batched trimmed ICP on the GPU (csrc/icp.hip) through the C ABI vs the CPU restatement of the
same definition (oracle/pgp_oracle.c:orc_icp).  PCL is not vendored in the reference, so parity
is against our own oracle: transforms within 2e-6, iteration counts equal,
energies within 1e-4 relative; plus invariants of the algorithm itself."""
import numpy as np
import pytest

from physimglobalpose_amd import LcpScorer, synth
from _checkers import oracle_icp

pytestmark = pytest.mark.gpu


def _problem(seed, n_model, n_seg, n_guess, rot_deg=8.0, trans=0.01, noise=0.0005):
    rng = np.random.default_rng(seed)
    M, _ = synth.make_model(rng, n_model)
    M = M.astype(np.float32)
    R = synth._random_rot(rng)
    t = rng.uniform(-0.2, 0.2, 3) + np.array([0, 0, 0.8])
    seg = M[rng.choice(n_model, n_seg, replace=False)]
    S = (seg @ R.T + t + noise * rng.standard_normal(seg.shape)).astype(np.float32)
    Tinv = np.linalg.inv(synth._se3(R, t))          # scene -> model frame (UCTState.cpp:184-185)
    G = np.stack([synth.colmajor16(Tinv @ synth._se3(synth._random_rot(rng, np.deg2rad(rot_deg)),
                                                     trans * rng.standard_normal(3)))
                  for _ in range(n_guess)])
    return S, M, G, Tinv


@pytest.mark.parametrize("trim,cap", [(0.9, 0.0), (1.0, 0.0), (0.5, 0.0), (1.0, 0.02)])
def test_matches_oracle(trim, cap):
    S, M, G, _ = _problem(1, 1500, 700, 12)
    sc = LcpScorer()
    T, e, it = sc.icp_refine(S, M, G, trim=trim, max_iterations=60, max_corr_dist=cap)
    To, eo, ito = oracle_icp(S, M, G, trim=trim, max_iterations=60, max_corr_dist=cap)
    assert np.array_equal(it, ito), (it, ito)
    assert np.abs(T - To).max() < 2e-6
    assert np.allclose(e, eo, rtol=1e-4, atol=1e-12)


def test_multi_tile_target_and_multi_sweep_source():
    """|tgt| > 4096 (two LDS tiles) and |src| > 4096 (two source sweeps per workgroup)."""
    S, M, G, Tinv = _problem(2, 6000, 4500, 3, rot_deg=4.0, trans=0.004)
    sc = LcpScorer()
    T, e, it = sc.icp_refine(S, M, G, trim=0.8, max_iterations=8)
    To, eo, ito = oracle_icp(S, M, G, trim=0.8, max_iterations=8)
    assert np.array_equal(it, ito) and np.abs(T - To).max() < 2e-6


def test_recovers_known_perturbation_and_energy_decreases():
    S, M, G, Tinv = _problem(3, 2500, 1200, 32, rot_deg=6.0, trans=0.008, noise=0.0003)
    sc = LcpScorer()
    T1, e1, it1 = sc.icp_refine(S, M, G, trim=0.9, max_iterations=1)
    T, e, it = sc.icp_refine(S, M, G, trim=0.9, max_iterations=100)
    assert (e <= e1 + 1e-12).all() and (it >= 2).all() and (it <= 100).all()
    err = np.abs(T.reshape(-1, 4, 4).transpose(0, 2, 1) - Tinv).max(axis=(1, 2))
    assert np.median(err) < 1e-3 and (err < 5e-3).mean() > 0.9
    # refined transforms are rigid
    Rm = T.reshape(-1, 4, 4).transpose(0, 2, 1)[:, :3, :3]
    assert np.abs(Rm @ Rm.transpose(0, 2, 1) - np.eye(3)).max() < 1e-5
    assert np.abs(np.linalg.det(Rm) - 1).max() < 1e-5


def test_identity_guess_on_identical_clouds_is_a_fixed_point():
    rng = np.random.default_rng(4)
    M = rng.uniform(-0.1, 0.1, (800, 3)).astype(np.float32)
    sc = LcpScorer()
    T, e, it = sc.icp_refine(M, M, synth.colmajor16(np.eye(4))[None], trim=1.0)
    assert e[0] == 0.0 and np.abs(T[0] - synth.colmajor16(np.eye(4))).max() < 1e-6


def test_bad_arguments():
    from physimglobalpose_amd import _lib
    sc = LcpScorer()
    with pytest.raises(_lib.PgpError):
        sc.icp_refine(np.zeros((0, 3), np.float32), np.zeros((5, 3), np.float32),
                      synth.colmajor16(np.eye(4))[None])
    T, e, it = sc.icp_refine(np.zeros((5, 3), np.float32), np.zeros((5, 3), np.float32),
                             np.zeros((0, 16), np.float32))
    assert len(T) == 0


@pytest.mark.parametrize("trim,cap", [(0.9, 0.0), (1.0, 0.015)])
def test_split_path_is_bit_identical_to_persistent_path(trim, cap, monkeypatch):
    """Few poses: correspondences searched by many workgroups per pose (icp_nn_split + 64-bit
    atomic-min keys); many poses: one persistent workgroup per pose.  Same result, bit for bit."""
    S, M, G, _ = _problem(5, 3000, 1700, 6)
    sc = LcpScorer()
    monkeypatch.setenv("PGP_ICP_SPLIT", "0")
    T0, e0, it0 = sc.icp_refine(S, M, G, trim=trim, max_iterations=40, max_corr_dist=cap)
    monkeypatch.setenv("PGP_ICP_SPLIT", "1")
    T1, e1, it1 = sc.icp_refine(S, M, G, trim=trim, max_iterations=40, max_corr_dist=cap)
    assert np.array_equal(it0, it1) and np.array_equal(T0, T1) and np.array_equal(e0, e1)
    To, eo, ito = oracle_icp(S, M, G, trim=trim, max_iterations=40, max_corr_dist=cap)
    assert np.array_equal(it1, ito) and np.abs(T1 - To).max() < 2e-6
