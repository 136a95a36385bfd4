"""Base selection on the device (csrc/base_select.hip) against the Eigen-typed harness fixture
tests/golden/stocs.npz: point-pair features, table look-ups, the three weighting loops of
SelectQuadrilateralStoCS (bit-exact, incl. the normalisation by the sequential float sum),
TryQuadrilateral, and the one-launch selection of many bases (draws checked against an inverse-CDF
emulation on the verified stage weights)."""
import os

import numpy as np
import pytest

from physimglobalpose_amd import LcpScorer

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stocs.npz")


@pytest.fixture(scope="module")
def ctx():
    g = np.load(GOLD)
    sc = LcpScorer()
    sc.set_scene(g["P"], g["N"], g["prob"], 0.005)
    sc.set_ppf_map(g["keys"])
    return g, sc


def test_point_pair_features_and_table_lookup(ctx):
    g, sc = ctx
    f, rows = sc.ppf_features(g["pairs"])
    assert np.array_equal(f, g["feat"])
    table = {tuple(k): i for i, k in enumerate(g["keys"].tolist())}
    want = np.array([table.get(tuple(x), -1) for x in g["feat"].tolist()], np.int32)
    assert np.array_equal(rows, want) and (want >= 0).any() and (want < 0).any()


def test_stage_weights_bit_exact(ctx):
    g, sc = ctx
    for c in range(6):
        b, s, present = g[f"b_{c}"], g[f"s_{c}"], g[f"present_{c}"]
        cur, sm, pr = sc.stocs_stage_weights(2, g["prob"], b[0])
        assert pr == bool(present[0]) and np.float32(sm) == s[0] and np.array_equal(cur, g[f"cur2_{c}"])
        cur, sm, pr = sc.stocs_stage_weights(3, cur, b[0], b[1])
        assert pr == bool(present[1]) and np.float32(sm) == s[1] and np.array_equal(cur, g[f"cur3_{c}"])
        if b[2] >= 0:
            cur, sm, pr = sc.stocs_stage_weights(4, cur, b[0], b[1], b[2])
            assert pr == bool(present[2]) and np.float32(sm) == s[2]
            assert np.array_equal(cur, g[f"cur4_{c}"])


def test_try_quadrilateral_bit_exact(ctx):
    g, sc = ctx
    ids, inv, ok = sc.base_invariants(g["quads"])
    assert np.array_equal(ok == 1, g["quad_ok"] == 1)
    good = g["quad_ok"] == 1
    assert np.array_equal(ids[good], g["quad_ids"][good])
    assert np.array_equal(inv[good], g["quad_inv"][good])


def _draw(w, u):
    c = np.cumsum(w.astype(np.float64))
    t = u * c[-1]
    i = int(np.searchsorted(c, t, side="left"))
    while w[i] == 0:
        i += 1
    margin = min(abs(c[i] - t), abs(t - (c[i - 1] if i else 0.0))) / c[-1]
    return i, margin


def test_select_bases_one_launch(ctx):
    g, sc = ctx
    rng = np.random.default_rng(7)
    u = rng.random((160, 4))
    ids, inv, status = sc.select_bases(u)
    assert set(np.unique(status)) <= {0, 1} and status.sum() >= 20
    checked = 0
    for a in range(len(u)):
        b1, m1 = _draw(g["prob"], u[a, 0])
        cur, _, p2 = sc.stocs_stage_weights(2, g["prob"], b1)
        margins, bs = [m1], [b1]
        ok = p2
        for stage in (3, 4):
            if not ok:
                break
            b, m = _draw(cur, u[a, stage - 2])
            bs.append(b)
            margins.append(m)
            cur, _, ok = sc.stocs_stage_weights(stage, cur, *bs)
        if min(margins) < 1e-9:          # a variate on a CDF boundary: the two prefix-sum associations may differ
            continue
        assert status[a] == int(ok), (a, bs)
        if not ok:
            continue
        b4, m4 = _draw(cur, u[a, 3])
        if m4 < 1e-9:
            continue
        want_ids, want_inv, _ = sc.base_invariants(np.array([bs + [b4]]))
        assert np.array_equal(ids[a], want_ids[0]) and np.array_equal(inv[a], want_inv[0])
        assert len(set(ids[a].tolist())) == 4 and (g["prob"][ids[a]] > 0).all()
        checked += 1
    assert checked >= 20
