"""world_size-2 (and 3) gloo runs of the multi-GPU sharding path on CPU: the sharded batch must
give the same score vector and the same best index as the unsharded one (the oracle stands in
for the per-rank device scorer -- checker use only)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from physimglobalpose_amd import synth
from physimglobalpose_amd.sharding import (MultiObjectShardedScorer, ShardedScorer, best_of, flat_slices,
                                           shard_bounds)


def test_shard_bounds_cover_and_balance():
    for n in (0, 1, 7, 4096, 65536, 65537):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_best_of_rule():
    assert best_of(torch.tensor([0.0, 0.0])) == (-1, 0.0)
    assert best_of(torch.tensor([])) == (-1, 0.0)
    bi, bs = best_of(torch.tensor([0.1, 0.5, 0.5, 0.2]))
    assert bi == 1 and abs(bs - 0.5) < 1e-7
    assert best_of(torch.tensor([float("nan"), 0.25]))[0] == 1


def test_flat_slices_partition_every_object_exactly_once():
    for counts in ([5], [3, 0, 9], [16384] * 3, [10923] * 6, [1, 1, 1, 1, 1, 1]):
        for world in (1, 2, 3, 8):
            seen = [np.zeros(c, int) for c in counts]
            sizes = []
            for r in range(world):
                n = 0
                for obj, lo, hi in flat_slices(counts, r, world):
                    seen[obj][lo:hi] += 1
                    n += hi - lo
                sizes.append(n)
            assert all((s == 1).all() for s in seen)
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_h, q):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here]
    from _checkers import Oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w = synth.make_workload(1500, 200, n_h, config_id=31)
        orc = Oracle(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm)

        def local(Ts):
            s, _, _ = orc.score_batch(Ts.numpy(), w.delta, mode=0)
            return torch.from_numpy(s)

        sh = ShardedScorer(local)
        scores, bi, bs = sh.score(torch.from_numpy(w.T))
        # several objects at once (configs[3] pattern): here the same scorer under 3 list lengths
        Ts = [torch.from_numpy(w.T[:n]) for n in (n_h, n_h // 3, 5)]
        per_obj, bests = MultiObjectShardedScorer([local] * 3).score(Ts)
        q.put((rank, scores.numpy(), bi, bs, [p.numpy() for p in per_obj], bests))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_h", [(2, 37), (3, 64)])
def test_sharded_equals_unsharded(world, n_h):
    from _checkers import Oracle
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_h, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    w = synth.make_workload(1500, 200, n_h, config_id=31)
    orc = Oracle(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm)
    s_ref, bi_ref, _ = orc.score_batch(w.T, w.delta, mode=0)
    for rank, s, bi, bs, per_obj, bests in results:
        assert np.array_equal(s, s_ref), f"rank {rank}"
        assert bi == bi_ref and np.float32(bs) == s_ref[bi_ref]
        for n, p, (obi, obs) in zip((n_h, n_h // 3, 5), per_obj, bests):
            assert np.array_equal(p, s_ref[:n])
            exp = int(np.argmax(s_ref[:n])) if s_ref[:n].max() > 0 else -1
            assert obi == exp


def _bucket_worker(rank, world, port, n_local, n_batches, bucket, q):
    from physimglobalpose_amd.sharding import BucketedExchange
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ex = BucketedExchange(n_local, rank, world, "cpu", bucket=bucket)
        seen = []
        for b in range(n_batches):
            g = torch.Generator().manual_seed(1000 * b + rank)
            ex.slot().copy_(torch.rand(n_local, generator=g))    # "scoring" this rank's slice of batch b
            ex.commit()
            if ex.argmax is not None:
                seen.append(ex.argmax.clone())
        ex.drain()
        q.put((rank, ex.last_vector().numpy().copy(), ex.argmax.numpy().copy(), len(seen)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bucket,n_batches", [(1, 3), (4, 10), (8, 8)])
def test_bucketed_exchange_gathers_every_batch(bucket, n_batches):
    """The pipelined exchange bench.py times (two alternating buckets, one all-reduce per bucket):
    after drain() the last batch's vector holds every rank's slice and the arg-max is the global one."""
    world, n_local = 2, 33
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, n_local, n_batches, bucket, q))
             for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    b = n_batches - 1
    want = torch.cat([torch.rand(n_local, generator=torch.Generator().manual_seed(1000 * b + r))
                      for r in range(world)]).numpy()
    for rank, vec, am, n_seen in results:
        assert np.array_equal(vec, want), f"rank {rank}"
        j = b % bucket
        assert int(am[j]) == int(np.argmax(want))


def test_native_flat_slices_equal_the_python_twin():
    """pgp_multi_slice / pgp_multi_flat_slices (csrc/multi_gpu.hip: how the C-ABI group partitions the flat (object,
    hypothesis), (job, pose) and base spaces) are pure host helpers: the same pieces as sharding.shard_bounds / flat_slices,
    every unit covered exactly once, without a GPU."""
    import ctypes as C
    from physimglobalpose_amd import _lib
    from physimglobalpose_amd.sharding import flat_slices, shard_bounds
    L = _lib.load()
    rng = np.random.default_rng(3)
    cases = [[16384, 12288, 12288, 8192, 8192, 8192], [5, 0, 3], [0, 0], [1], [7, 1, 1, 1, 90]]
    cases += [rng.integers(0, 50, rng.integers(1, 9)).tolist() for _ in range(40)]
    for counts in cases:
        n = len(counts)
        arr = (C.c_int * n)(*counts)
        for world in (1, 2, 3, 5, 8):
            seen = []
            for k in range(world):
                lo, hi = C.c_int(), C.c_int()
                assert L.pgp_multi_slice(sum(counts), k, world, C.byref(lo), C.byref(hi)) == 0
                assert (lo.value, hi.value) == shard_bounds(sum(counts), k, world)
                o, a, b = (C.c_int * n)(), (C.c_int * n)(), (C.c_int * n)()
                m = C.c_int()
                assert L.pgp_multi_flat_slices(arr, n, k, world, o, a, b, C.byref(m)) == 0
                got = [(o[i], a[i], b[i]) for i in range(m.value)]
                assert got == flat_slices(counts, k, world)
                seen += [(obj, i) for obj, x, y in got for i in range(x, y)]
            assert seen == [(obj, i) for obj, c in enumerate(counts) for i in range(c)]
    assert L.pgp_multi_slice(10, 3, 3, C.byref(C.c_int()), C.byref(C.c_int())) == -1      # member index out of range
