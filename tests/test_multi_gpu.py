"""The native multi-GPU group behind the C ABI (pgp_multi_*, csrc/multi_gpu.hip): slice arithmetic
on the CPU; on the GPU box a one-device group -- with and without the RCCL exchange
(PGP_MULTI_FORCE_COLLECTIVE=1 runs a one-rank communicator: ncclCommInitAll, ONE grouped integer
all-reduce over {scores | counts}, then the arg-max over the complete vector on device 0) -- returns what a single
context returns, bit for bit."""
import os

import numpy as np
import pytest

from physimglobalpose_amd import LcpScorer, MultiGpuScorer, PGP_MODE_PLAIN, PGP_MODE_WEIGHTED, synth
from physimglobalpose_amd.sharding import shard_bounds

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_slices_partition_the_batch_like_the_torch_side():
    for n_total in (0, 1, 7, 8, 9, 4096, 65536, 65537):
        for n_dev in (1, 2, 3, 8):
            cover = []
            for k in range(n_dev):
                lo, hi = MultiGpuScorer.slice_of(n_total, k, n_dev)
                assert (lo, hi) == shard_bounds(n_total, k, n_dev)
                cover.append((lo, hi))
            assert cover[0][0] == 0 and cover[-1][1] == n_total
            assert all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
            sizes = [b - a for a, b in cover]
            assert max(sizes) - min(sizes) <= 1


def test_slice_rejects_bad_arguments():
    import ctypes as C
    from physimglobalpose_amd import _lib
    L = _lib.load()
    lo, hi = C.c_int(), C.c_int()
    assert L.pgp_multi_slice(10, 3, 3, C.byref(lo), C.byref(hi)) == -1
    assert L.pgp_multi_slice(-1, 0, 3, C.byref(lo), C.byref(hi)) == -1
    assert L.pgp_multi_slice(10, 0, 0, C.byref(lo), C.byref(hi)) == -1


@pytest.mark.gpu
@pytest.mark.parametrize("force", ["0", "1"])
def test_one_device_group_equals_single_context(force, monkeypatch):
    monkeypatch.setenv("PGP_MULTI_FORCE_COLLECTIVE", force)
    w = synth.make_workload(20000, 2000, 777, config_id=41)
    one = LcpScorer(0)
    one.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    grp = MultiGpuScorer([0])
    assert grp.n_devices == 1
    grp.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    for mode in (PGP_MODE_PLAIN, PGP_MODE_WEIGHTED):
        a = one.score(w.T, mode, w.gate_deg)
        b = grp.score(w.T, mode, w.gate_deg)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2:] == b[2:]
        grp.upload(w.T[:100])
        c = grp.score_uploaded(mode, w.gate_deg)
        d = one.score(w.T[:100], mode, w.gate_deg)
        assert np.array_equal(c[0], d[0]) and c[2:] == d[2:]
    s, c, bi, bs = grp.score(np.zeros((0, 16), np.float32), PGP_MODE_WEIGHTED)
    assert len(s) == 0 and bi == -1 and bs == 0.0
    t = grp.last_timing()
    assert t["total_ms"] >= 0
    grp.close()


@pytest.mark.gpu
def test_group_settles_near_ties_like_the_reference(monkeypatch):
    """The arg-max over the assembled vector goes through pgp_settle_best_device: the near-tie
    fixture's best index is the reference's."""
    monkeypatch.setenv("PGP_MULTI_FORCE_COLLECTIVE", "1")
    g = np.load(os.path.join(GOLD, "near_ties.npz"))
    grp = MultiGpuScorer([0])
    grp.init(g["P"], g["Pn"], g["Pw"], g["Q"], g["Qn"], float(g["delta"]))
    s, c, bi, bs = grp.score(g["T"], PGP_MODE_WEIGHTED, 30.0)
    assert bi == int(g["best_weighted"]) and np.float32(bs) == g["wscores"][bi]
    assert np.allclose(s, g["wscores"], rtol=0, atol=2e-6)


@pytest.mark.gpu
def test_settle_best_over_an_assembled_vector():
    """Scores produced by two separate device calls (two 'slices'), then ONE settle over the whole
    vector: same best as a single call over the whole batch -- what sharding.py does after its
    all-reduce."""
    import torch
    g = np.load(os.path.join(GOLD, "near_ties.npz"))
    sc = LcpScorer(0)
    sc.init(g["P"], g["Pn"], g["Pw"], g["Q"], g["Qn"], float(g["delta"]))
    n = len(g["T"])
    sc.reserve(n)
    dT = torch.from_numpy(g["T"]).cuda()
    ds = torch.zeros(n, device="cuda")
    cut = 100   # the near-tie cluster straddles the cut
    assert (g["cluster"] < cut).any() and (g["cluster"] >= cut).any()
    sc.score_device(dT[:cut], ds[:cut], mode=PGP_MODE_WEIGHTED)
    sc.score_device(dT[cut:], ds[cut:], mode=PGP_MODE_WEIGHTED)
    db = torch.zeros(2, dtype=torch.int32, device="cuda")
    sc.settle_best_device(dT, ds, db, mode=PGP_MODE_WEIGHTED)
    torch.cuda.synchronize()
    assert int(db[0]) == int(g["best_weighted"])
    assert np.float32(db[1:].view(torch.float32).item()) == g["wscores"][int(db[0])]
