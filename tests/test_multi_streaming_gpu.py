"""The streaming form of the device group (pgp_multi_upload_slot / _enqueue_slot / _collect, csrc/multi_gpu.hip): resident
batches, steps queued without a host wait, the exchange of step i on a second stream under the scoring of step i + 1, member
0's arg-max behind that scoring.  Whatever the interleaving, the LAST step's arrays are those of one context scoring the same
list (base.cc:1885-1901 over successive lists).  Run as n emulated members on ONE device (PGP_MULTI_EMULATE), as one member
with a real one-rank RCCL communicator (PGP_MULTI_FORCE_COLLECTIVE), and as the one-rank case of a group that spans
processes (pgp_multi_create_ranked: ncclGetUniqueId + ncclCommInitRank)."""
import os

import numpy as np
import pytest

from physimglobalpose_amd import LcpScorer, MultiGpuScorer, PGP_MODE_PLAIN, PGP_MODE_WEIGHTED, synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _same(a, b):
    return np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2:] == b[2:]


def _workload():
    w = synth.make_workload(20000, 2000, 4 * 600, config_id=43)
    return w, [w.T[k * 600:(k + 1) * 600] for k in range(4)]


def _check_streaming(grp, one, w, lists):
    grp.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    lists = lists + [w.T[:5], w.T[:0]]            # fewer hypotheses than members; the empty list
    for k, T in enumerate(lists):
        grp.upload_slot(k, T)
    for mode in (PGP_MODE_PLAIN, PGP_MODE_WEIGHTED):
        want = [one.score(T, mode, w.gate_deg) for T in lists]
        # one step, two steps (both rings), many steps ending on every slot
        for order in ([0], [1, 2], [0, 1, 2, 3, 0, 1, 2], [3, 2, 1, 0, 4], [0, 5], [5, 1], [4, 4, 4], [2, 5, 5]):
            for s in order:
                grp.enqueue_slot(s, mode, w.gate_deg)
            got = grp.collect()
            assert _same(got, want[order[-1]]), (mode, order)
    # a step queued and never collected before the batches change is refused, not silently dropped
    grp.enqueue_slot(0, PGP_MODE_PLAIN)
    with pytest.raises(Exception, match="in flight"):
        grp.upload_slot(0, lists[1])
    assert _same(grp.collect(), one.score(lists[0], PGP_MODE_PLAIN))
    with pytest.raises(Exception, match="no step"):
        grp.collect()
    # the synchronous calls still work beside it
    assert _same(grp.score(lists[2], PGP_MODE_WEIGHTED, w.gate_deg), one.score(lists[2], PGP_MODE_WEIGHTED, w.gate_deg))


@pytest.mark.parametrize("n", [2, 3, 8])
def test_streaming_emulated_members(n, monkeypatch):
    w, lists = _workload()
    one = LcpScorer(0)
    one.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    monkeypatch.setenv("PGP_MULTI_EMULATE", str(n))
    grp = MultiGpuScorer([0])
    inf = grp.info()
    assert inf["world"] == n and inf["n_local"] == n and inf["emulated"] and inf["rccl_ranks"] == 0
    _check_streaming(grp, one, w, lists)
    assert grp.info()["exchanges"] > 0
    grp.close()


def test_streaming_one_member_no_collective():
    w, lists = _workload()
    one = LcpScorer(0)
    one.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    grp = MultiGpuScorer([0])
    assert grp.info()["rccl_ranks"] == 0 and grp.info()["world"] == 1
    _check_streaming(grp, one, w, lists)
    assert grp.info()["exchanges"] == 0
    grp.close()


def test_streaming_one_rank_rccl(monkeypatch):
    """the real all-reduce on the second stream, one rank: ncclCommInitAll"""
    w, lists = _workload()
    one = LcpScorer(0)
    one.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    monkeypatch.setenv("PGP_MULTI_FORCE_COLLECTIVE", "1")
    grp = MultiGpuScorer([0])
    assert grp.info()["rccl_ranks"] == 1
    _check_streaming(grp, one, w, lists)
    assert grp.info()["exchanges"] > 0
    grp.close()


def test_ranked_group_of_one_process(monkeypatch):
    """pgp_multi_unique_id + pgp_multi_create_ranked (ncclCommInitRank), world 1: the multi-process form's own code"""
    w, lists = _workload()
    one = LcpScorer(0)
    one.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    monkeypatch.setenv("PGP_MULTI_FORCE_COLLECTIVE", "1")
    uid = MultiGpuScorer.unique_id()
    assert len(uid) == 128 and any(uid)
    grp = MultiGpuScorer.ranked([0], 0, 1, uid)
    inf = grp.info()
    assert inf["rccl_ranks"] == 1 and inf["world"] == 1 and inf["rank0"] == 0 and inf["devices"] == [0]
    _check_streaming(grp, one, w, lists)
    grp.close()
    with pytest.raises(Exception, match="ranks"):
        MultiGpuScorer.ranked([0], 1, 1, uid)       # rank 1 of a world of 1


@pytest.mark.parametrize("n", [2, 8])
def test_streaming_settles_near_ties_and_records_across_slices(n, monkeypatch):
    g = np.load(os.path.join(GOLD, "near_ties.npz"))
    monkeypatch.setenv("PGP_MULTI_EMULATE", str(n))
    grp = MultiGpuScorer([0])
    grp.init(g["P"], g["Pn"], g["Pw"], g["Q"], g["Qn"], float(g["delta"]))
    grp.upload_slot(0, g["T"])
    grp.upload_slot(1, g["T"][::-1].copy())
    for s in (1, 0, 1, 0):
        grp.enqueue_slot(s, PGP_MODE_WEIGHTED, 30.0)
    sc, c, bi, bs = grp.collect()
    assert bi == int(g["best_weighted"]) and np.float32(bs) == g["wscores"][bi]
    assert np.allclose(sc, g["wscores"], rtol=0, atol=2e-6)
    # exact records over the complete vector (set on member 0's context), through the streaming tail
    one = LcpScorer(0)
    one.init(g["P"], g["Pn"], g["Pw"], g["Q"], g["Qn"], float(g["delta"]))
    one.set_exact_records(True)
    grp.set_exact_records(True)
    a = one.score(g["T"], PGP_MODE_WEIGHTED, 30.0)
    for s in (1, 0):
        grp.enqueue_slot(s, PGP_MODE_WEIGHTED, 30.0)
    assert _same(grp.collect(), a)
    grp.close()
