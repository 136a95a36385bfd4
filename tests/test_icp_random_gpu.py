"""Randomised ICP problems: target and source sizes from a handful to thousands, guesses from converged to far off,
trim fractions / caps / both metrics, few and many poses -- the default path (index, clustered launch where it
applies) must return the bits of the exhaustive scan (UCTState.cpp:121-204 call form and its siblings)."""
import numpy as np
import pytest

from physimglobalpose_amd import LcpScorer, synth

pytestmark = pytest.mark.gpu


def _case(seed):
    rng = np.random.default_rng(5000 + seed)
    n_tgt = int(rng.choice([5, 40, 300, 1500, 5000, 6500]))
    n_src = int(rng.choice([3, 64, 200, 1100, 2500, 4096, 4500]))
    n_pose = int(rng.choice([1, 3, 17, 64, 70, 130]))
    M, N = synth.make_model(rng, n_tgt)
    M = (M * float(rng.uniform(0.3, 3.0))).astype(np.float32)
    R = synth._random_rot(rng)
    t = rng.uniform(-0.5, 0.5, 3)
    S = (M[rng.integers(0, n_tgt, n_src)] @ R.T + t + 0.0005 * rng.standard_normal((n_src, 3)))
    if seed % 4 == 0:
        S[rng.integers(0, n_src, max(1, n_src // 10))] += rng.uniform(-0.2, 0.2, 3)
    S = S.astype(np.float32)
    Tinv = np.linalg.inv(synth._se3(R, t))
    rot = float(rng.choice([0.2, 3.0, 12.0]))
    G = np.stack([synth.colmajor16(Tinv @ synth._se3(synth._random_rot(rng, np.deg2rad(rot)), 0.002 * rot * rng.standard_normal(3)))
                  for _ in range(n_pose)])
    form = [dict(max_iterations=12, trim_fraction=0.9, energy_ratio=1.0),
            dict(max_iterations=9, trim_fraction=0.6, energy_ratio=0.0),
            dict(max_iterations=10, max_corr_dist=0.03, energy_ratio=0.0, transformation_epsilon=1e-9, absolute_mse=1e-14),
            dict(max_iterations=8, energy_ratio=0.0, error_metric=1, transformation_epsilon=0.0, absolute_mse=1e-14),
            dict(max_iterations=10, trim_fraction=1.0, energy_ratio=1.0)][seed % 5]
    return S, M, N.astype(np.float32), G, form


@pytest.mark.parametrize("seed", range(20))
def test_random_icp_default_path_equals_scan(seed, monkeypatch):
    S, M, N, G, form = _case(seed)
    sc = LcpScorer()
    nrm = N if form.get("error_metric") == 1 else None
    monkeypatch.setenv("PGP_ICP_NN", "scan")
    ref = sc.icp_refine_ex(S, M, G, tgt_nrm=nrm, **form)
    monkeypatch.delenv("PGP_ICP_NN")
    got = sc.icp_refine_ex(S, M, G, tgt_nrm=nrm, **form)
    for x, y, what in zip(ref, got, ("T", "energy", "iters")):
        assert np.array_equal(x, y), (seed, what, len(S), len(M), len(G))
