"""All bases of an object in one pass (pgp_find_congruent_batch: pair lists looked up in the device
pair-feature table, one bucket table and one sort for every base) against the single-base entry
point (which is pinned on the reference's own IndexedNormalSet, tests/golden/congruent_*.npz):
identical quad lists, in the same order, and identical rigid fits for picked (base, quad) pairs."""
import tempfile

import numpy as np
import pytest

from physimglobalpose_amd import LcpScorer
from _dropin import make_dropin_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def case():
    with tempfile.TemporaryDirectory() as d:
        _, c = make_dropin_case(d, n_scene=8000, n_model=1500, n_search=500)
    w, table = c["w"], c["table"]
    keys = np.array(list(table.keys()), np.int32)
    counts = np.array([len(table[tuple(k)]) for k in keys.tolist()], np.int32)
    pairs = np.concatenate([np.array(table[tuple(k)], np.int32).reshape(-1, 2) for k in keys.tolist()])
    sc = LcpScorer()
    sc.set_scene(w.P_xyz, w.P_nrm, w.P_w, w.delta)
    sc.set_search_model(w.Qs_xyz)
    sc.set_ppf_map(keys, counts, pairs)
    rng = np.random.default_rng(3)
    ids, inv, status = sc.select_bases(rng.random((96, 4)))
    ok = status == 1
    return w, table, keys, sc, ids[ok], inv[ok]


def test_batch_equals_base_by_base(case):
    w, table, keys, sc, ids, inv = case
    assert len(ids) >= 16
    base_xyz = w.P_xyz[ids]                                    # (nb, 4, 3)
    n_quads = sc.find_congruent_batch(ids, base_xyz, inv, w.delta)
    f01, r01 = sc.ppf_features(ids[:, [0, 1]])
    f23, r23 = sc.ppf_features(ids[:, [2, 3]])
    total = 0
    per_base = []
    for b in range(len(ids)):
        if r01[b] < 0 or r23[b] < 0:
            assert n_quads[b] == 0
            per_base.append(np.zeros((0, 4), np.int32))
            continue
        p1 = np.array(table[tuple(keys[r01[b]].tolist())], np.int32)
        p6 = np.array(table[tuple(keys[r23[b]].tolist())], np.int32)
        q = sc.find_congruent(base_xyz[b], inv[b, 0], inv[b, 1], w.delta, p1, p6)
        assert n_quads[b] == len(q), b
        per_base.append(q)
        total += len(q)
    assert total > 0
    # the batch's resident lists: every quad of every base, in the reference's order
    sc.find_congruent_batch(ids, base_xyz, inv, w.delta)       # the single-base calls reused the workspaces
    picks = np.array([(b, j) for b in range(len(ids)) for j in range(n_quads[b])], np.int32).reshape(-1, 2)
    got = sc.congruent_batch_quads(picks)
    assert np.array_equal(got, np.concatenate(per_base))
    # rigid fits of a sample of picks, without the quads leaving the device
    rng = np.random.default_rng(5)
    sel = picks[rng.choice(len(picks), min(len(picks), 2000), replace=False)]
    T, pose, status, rms = sc.congruent_batch_fit(sel, ids, w.centroid_P, w.centroid_Q)
    quads_sel = np.array([per_base[b][j] for b, j in sel], np.int32)
    T2, pose2, status2, rms2 = sc.rigid_from_congruent(ids[sel[:, 0]], quads_sel, w.centroid_P, w.centroid_Q)
    assert np.array_equal(status, status2) and (status == 1).any()
    good = status == 1
    assert np.array_equal(T[good], T2[good]) and np.array_equal(pose[good], pose2[good]) and np.array_equal(rms, rms2)


def test_batch_without_pair_lists_is_refused(case):
    w, table, keys, sc, ids, inv = case
    other = LcpScorer()
    other.set_scene(w.P_xyz, w.P_nrm, w.P_w, w.delta)
    other.set_search_model(w.Qs_xyz)
    other.set_ppf_map(keys)                                     # presence only
    from physimglobalpose_amd._lib import PgpError
    with pytest.raises(PgpError):
        other.find_congruent_batch(ids, w.P_xyz[ids], inv, w.delta)


def test_stale_batches_and_bad_picks_are_refused(case):
    """The resident quad lists die with the buffers they index: a single-base pgp_find_congruent, a new
    pair-feature table or a new search model discard them (PGP_ESTATE instead of quads gathered from
    overwritten keys), and a pick outside a base's list is PGP_EINVAL, not a device read past the end."""
    w, table, keys, sc, ids, inv = case
    from physimglobalpose_amd._lib import PgpError
    base_xyz = w.P_xyz[ids]
    n_quads = sc.find_congruent_batch(ids, base_xyz, inv, w.delta)
    b = int(np.argmax(n_quads))
    assert n_quads[b] > 0
    ok = sc.congruent_batch_quads(np.array([[b, 0], [b, n_quads[b] - 1]], np.int32))
    assert ok.shape == (2, 4)
    for bad in ([[b, n_quads[b]]], [[b, -1]], [[len(ids), 0]], [[-1, 0]]):
        with pytest.raises(PgpError):
            sc.congruent_batch_quads(np.array(bad, np.int32))
        with pytest.raises(PgpError):
            sc.congruent_batch_fit(np.array(bad, np.int32), ids, w.centroid_P, w.centroid_Q)
    # a single-base call overwrites the sorted keys
    f01, r01 = sc.ppf_features(ids[b:b + 1, [0, 1]])
    f23, r23 = sc.ppf_features(ids[b:b + 1, [2, 3]])
    p1 = np.array(table[tuple(keys[r01[0]].tolist())], np.int32)
    p6 = np.array(table[tuple(keys[r23[0]].tolist())], np.int32)
    sc.find_congruent(base_xyz[b], inv[b, 0], inv[b, 1], w.delta, p1, p6)
    with pytest.raises(PgpError):
        sc.congruent_batch_quads(np.array([[b, 0]], np.int32))
    # ... and so do a new table and a new search model
    counts = np.array([len(table[tuple(k)]) for k in keys.tolist()], np.int32)
    pairs = np.concatenate([np.array(table[tuple(k)], np.int32).reshape(-1, 2) for k in keys.tolist()])
    for invalidate in (lambda: sc.set_ppf_map(keys, counts, pairs), lambda: sc.set_search_model(w.Qs_xyz)):
        sc.find_congruent_batch(ids, base_xyz, inv, w.delta)
        assert sc.congruent_batch_quads(np.array([[b, 0]], np.int32)).shape == (1, 4)
        invalidate()
        with pytest.raises(PgpError):
            sc.congruent_batch_quads(np.array([[b, 0]], np.int32))
    sc.set_ppf_map(keys, counts, pairs)
    sc.find_congruent_batch(ids, base_xyz, inv, w.delta)
    assert sc.congruent_batch_quads(np.array([[b, 0]], np.int32)).shape == (1, 4)


def test_a_batch_that_outgrows_its_key_array_is_matched_again():
    """The batch's single matching pass appends its keys to an array sized by a guess; PGP_CS_KEY_CAP=8 makes every batch
    outgrow it, so the grow-and-repeat path runs: quad counts and quads equal the default run's."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import json, sys, tempfile, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from physimglobalpose_amd import LcpScorer
from _dropin import make_dropin_case
with tempfile.TemporaryDirectory() as d:
    _, c = make_dropin_case(d, n_scene=6000, n_model=1200, n_search=400)
w, table = c["w"], c["table"]
keys = np.array(list(table.keys()), np.int32)
counts = np.array([len(table[tuple(k)]) for k in keys.tolist()], np.int32)
pairs = np.concatenate([np.array(table[tuple(k)], np.int32).reshape(-1, 2) for k in keys.tolist()])
sc = LcpScorer()
sc.set_scene(w.P_xyz, w.P_nrm, w.P_w, w.delta)
sc.set_search_model(w.Qs_xyz)
sc.set_ppf_map(keys, counts, pairs)
ids, inv, status = sc.select_bases(np.random.default_rng(3).random((64, 4)))
ok = status == 1
ids, inv = ids[ok], inv[ok]
out = []
for rep in range(2):
    n_quads = sc.find_congruent_batch(ids, w.P_xyz[ids], inv, w.delta)
    picks = np.array([(b, j) for b in range(len(ids)) for j in range(n_quads[b])], np.int32).reshape(-1, 2)
    out.append([n_quads.tolist(), sc.congruent_batch_quads(picks).tolist()])
assert out[0] == out[1]
print("RESULT " + json.dumps(out[0]))
""" % (root, os.path.join(root, "tests"))
    res = []
    for env in ({}, {"PGP_CS_KEY_CAP": "8"}):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        res.append(json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:]))
    assert res[0] == res[1] and sum(res[0][0]) > 8



def test_rows_that_ride_home_with_the_bases(case):
    """pgp_select_bases_rows hands out the table rows of every base's two edges (what ExtractCongruentSet looks up,
    base.cc:1970-1981) and pgp_find_congruent_batch_rows takes them: same bases, same rows as pgp_ppf_features, same quads
    in the same order as the call that asks for the rows itself."""
    w, table, keys, sc, ids0, inv0 = case
    u = np.random.default_rng(3).random((96, 4))
    ids_a, inv_a, st_a = sc.select_bases(u)
    ids, inv, st, rows = sc.select_bases(u, rows=True)
    assert np.array_equal(ids, ids_a) and np.array_equal(inv, inv_a) and np.array_equal(st, st_a)
    assert (rows[st != 1] == -1).all()
    ok = st == 1
    ids, inv, rows = ids[ok], inv[ok], rows[ok]
    _, r01 = sc.ppf_features(ids[:, [0, 1]])
    _, r23 = sc.ppf_features(ids[:, [2, 3]])
    assert np.array_equal(rows[:, 0], r01) and np.array_equal(rows[:, 1], r23) and (rows >= 0).any()
    base_xyz = w.P_xyz[ids]
    n_plain = sc.find_congruent_batch(ids, base_xyz, inv, w.delta)
    picks = np.array([(b, j) for b in range(len(ids)) for j in range(n_plain[b])], np.int32).reshape(-1, 2)
    q_plain = sc.congruent_batch_quads(picks)
    # an ODD number of attempts: the 8-byte row records must not land on an odd word behind the 28-byte records
    for n_att in (95, 1, 33):
        i2, v2, s2, r2 = sc.select_bases(u[:n_att], rows=True)
        assert np.array_equal(i2, ids_a[:n_att]) and np.array_equal(v2, inv_a[:n_att]) and np.array_equal(s2, st_a[:n_att])
        assert np.array_equal(r2[s2 == 1], rows[:int((st_a[:n_att] == 1).sum())])
    # the selection in two halves (pgp_select_bases_rows_begin / _end) with a pgp_set_model in between: the same bases
    for n_att in (96, 33):
        sc.select_bases_begin(u[:n_att])
        sc.set_model(w.Q_xyz, w.Q_nrm)
        i3, v3, s3, r3 = sc.select_bases_end()
        assert np.array_equal(i3, ids_a[:n_att]) and np.array_equal(v3, inv_a[:n_att]) and np.array_equal(s3, st_a[:n_att])
        assert np.array_equal(r3[s3 == 1], rows[:int((st_a[:n_att] == 1).sum())])
    with pytest.raises(Exception, match="no selection"):
        sc.select_bases_end()
    n_rows = sc.find_congruent_batch(ids, base_xyz, inv, w.delta, rows=rows)
    assert np.array_equal(n_rows, n_plain) and n_plain.sum() > 0
    assert np.array_equal(sc.congruent_batch_quads(picks), q_plain)
    bad = rows.copy()
    bad[0, 0] = len(keys)          # one past the table
    with pytest.raises(Exception):
        sc.find_congruent_batch(ids, base_xyz, inv, w.delta, rows=bad)


def test_fit_score_list_equals_the_calls_it_replaces(case):
    """pgp_congruent_batch_fit_score_list (fits, verification, the running-best walk on the device, the kept poses, the best
    pose and its registered points in one call) against pgp_congruent_batch_fit_score -> pgp_running_best ->
    pgp_congruent_batch_fetch -> pgp_registered."""
    import ctypes as C
    from physimglobalpose_amd import PGP_MODE_WEIGHTED
    w, table, keys, sc0, ids, inv = case
    counts = np.array([len(table[tuple(k)]) for k in keys.tolist()], np.int32)
    pairs = np.concatenate([np.array(table[tuple(k)], np.int32).reshape(-1, 2) for k in keys.tolist()])
    sc = LcpScorer()
    sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    sc.set_search_model(w.Qs_xyz)
    sc.set_ppf_map(keys, counts, pairs)
    sc.set_exact_records(True)
    base_xyz = w.P_xyz[ids]
    n_quads = sc.find_congruent_batch(ids, base_xyz, inv, w.delta)
    rng = np.random.default_rng(11)
    picks = np.array([(b, j) for b in range(len(ids)) for j in rng.permutation(n_quads[b])[:40]], np.int32).reshape(-1, 2)
    m = len(picks)
    assert m > 200
    L, h = sc._lib, sc._h
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    cP, cQ = np.ascontiguousarray(w.centroid_P, np.float32), np.ascontiguousarray(w.centroid_Q, np.float32)
    ids_c = np.ascontiguousarray(ids, np.int32)
    scores, status = np.zeros(m, np.float32), np.zeros(m, np.int32)
    best, bscore = C.c_int(-1), C.c_float(0)
    assert L.pgp_congruent_batch_fit_score(h, ip(picks), ip(ids_c), m, fp(cP), fp(cQ), PGP_MODE_WEIGHTED, C.c_float(30.0), fp(scores),
                                           ip(status), C.byref(best), C.byref(bscore)) == 0
    sel = LcpScorer.running_best(scores)
    assert len(sel) >= 1 and best.value >= 0
    want = np.ascontiguousarray(list(sel) + [best.value], np.int32)
    T, pose = np.zeros((len(want), 16), np.float32), np.zeros((len(want), 16), np.float64)
    assert L.pgp_congruent_batch_fetch(h, ip(want), len(want), fp(T), dp(pose)) == 0
    reg = sc.registered(T[-1], PGP_MODE_WEIGHTED, 30.0)
    for cap in (256, 1, 0):
        got = sc.congruent_batch_fit_score_list(picks, ids_c, cP, cQ, PGP_MODE_WEIGHTED, 30.0, list_cap=cap)
        k = min(len(sel), cap)
        assert got["n_list"] == len(sel) and got["n_pushed"] == int((status == 1).sum())
        assert np.array_equal(got["index"], np.asarray(sel[:k], np.int32)) and np.array_equal(got["score"], scores[sel[:k]])
        assert np.array_equal(got["T"], T[:k]) and np.array_equal(got["pose"], pose[:k])
        assert got["best_index"] == best.value and got["best_score"] == float(np.float32(bscore.value))
        assert np.array_equal(got["best_T"], T[-1]) and np.array_equal(got["best_pose"], pose[-1])
        assert np.array_equal(got["registered"], reg) and len(reg) > 0
    # the resident fits still answer a fetch
    T2, pose2 = np.zeros_like(T), np.zeros_like(pose)
    assert L.pgp_congruent_batch_fetch(h, ip(want), len(want), fp(T2), dp(pose2)) == 0
    assert np.array_equal(T2, T) and np.array_equal(pose2, pose)
    sc.close()


def test_quads_drawn_on_the_device_equal_the_host_statement_and_the_picked_call(case):
    """pgp_congruent_batch_sample_fit_score_list draws the reference's sample of at most 100 quads per base (base.cc:1858-1866) ON
    THE DEVICE, one wave per base from a generator of the base's own: the picks it reports are pgp_sample_quads' (the same draw
    stated on the host), and everything it returns is what pgp_congruent_batch_fit_score_list returns for those picks."""
    from physimglobalpose_amd import PGP_MODE_WEIGHTED
    w, table, keys, sc0, ids, inv = case
    counts = np.array([len(table[tuple(k)]) for k in keys.tolist()], np.int32)
    pairs = np.concatenate([np.array(table[tuple(k)], np.int32).reshape(-1, 2) for k in keys.tolist()])
    sc = LcpScorer()
    sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    sc.set_search_model(w.Qs_xyz)
    sc.set_ppf_map(keys, counts, pairs)
    sc.set_exact_records(True)
    n_quads = sc.find_congruent_batch(ids, w.P_xyz[ids], inv, w.delta)
    assert (n_quads >= 100).any() and (n_quads < 100).any()          # both branches of the draw
    cP, cQ = np.ascontiguousarray(w.centroid_P, np.float32), np.ascontiguousarray(w.centroid_Q, np.float32)
    for seed, cap_q in ((12345, 100), (2 ** 63 + 7, 100), (99, 7), (5, 128), (6, 1)):
        picks = LcpScorer.sample_quads(seed, n_quads, cap_q)
        assert len(picks) == int(np.minimum(n_quads, cap_q).sum())
        for b in range(len(n_quads)):                                  # distinct, ascending, in range, all of a small base
            j = picks[picks[:, 0] == b, 1]
            assert len(j) == min(int(n_quads[b]), cap_q) and (np.diff(j) > 0).all() and (len(j) == 0 or (0 <= j[0] and j[-1] < n_quads[b]))
        got = sc.congruent_batch_sample_fit_score_list(seed, ids, cP, cQ, max_per_base=cap_q, list_cap=256)
        assert np.array_equal(got["picks"], picks), (seed, cap_q)
        n_quads2 = sc.find_congruent_batch(ids, w.P_xyz[ids], inv, w.delta)     # (the picked call on a batch of its own)
        assert np.array_equal(n_quads2, n_quads)
        ref = sc.congruent_batch_fit_score_list(picks, ids, cP, cQ, PGP_MODE_WEIGHTED, 30.0, list_cap=256)
        for k in ("n_list", "n_pushed", "best_index", "best_score"):
            assert got[k] == ref[k], (k, seed, cap_q)
        for k in ("index", "score", "T", "pose", "best_T", "best_pose", "registered"):
            assert np.array_equal(got[k], ref[k]), (k, seed, cap_q)
        sc.find_congruent_batch(ids, w.P_xyz[ids], inv, w.delta)
    # the draw of one base does not depend on the others (its generator is its own)
    a = LcpScorer.sample_quads(777, n_quads, 100)
    b = LcpScorer.sample_quads(777, n_quads[3:], 100)
    assert np.array_equal(a[a[:, 0] == 0, 1], LcpScorer.sample_quads(777, n_quads[:1], 100)[:, 1])
    assert not np.array_equal(a[a[:, 0] == 3, 1], b[b[:, 0] == 0, 1]) or n_quads[3] < 100      # base 3 as base 0 of another batch: another stream
    with pytest.raises(Exception, match="max_per_base"):
        sc.congruent_batch_sample_fit_score_list(1, ids, cP, cQ, max_per_base=129)
    sc.close()
