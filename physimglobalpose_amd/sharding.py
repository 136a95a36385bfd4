"""Multi-GPU sharding of the verification loop.

Hypotheses are independent, so the batch is block-partitioned over the ranks of one node (one
process per GPU, clouds + index replicated) and the only exchange is the score vector: each rank
fills its slice of a zero-initialised float[n_total] and one all-reduce(SUM) over RCCL/xGMI
(backend "nccl" on ROCm; "gloo" in the CPU tests) leaves every rank with all scores, as the host
MCTS needs them (HypothesisSelection.cpp:248-257).  The arg-max is then taken locally with the
reference's rule: lowest index of the maximum, -1 if the maximum is not > 0 (base.cc:1891,309).
The message is n_total*4 B (256 KiB at 64 k hypotheses): latency-bound, one collective per batch.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(n_total: int, rank: int, world: int):
    """Contiguous slice [lo, hi) of rank `rank`; sizes differ by at most one, earlier ranks larger."""
    base, rem = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def flat_slices(counts, rank: int, world: int):
    """Several objects, each with its own hypothesis list (BASELINE.json configs[3]: 6 objects,
    64 k hypotheses over 8 GPUs): flatten (object, hypothesis) into one index space, give rank
    `rank` its contiguous share, and return it as [(object, lo, hi), ...] pieces."""
    counts = [int(c) for c in counts]
    lo, hi = shard_bounds(sum(counts), rank, world)
    out, base = [], 0
    for obj, c in enumerate(counts):
        a, b = max(lo, base), min(hi, base + c)
        if b > a:
            out.append((obj, a - base, b - base))
        base += c
    return out


def best_of(scores: torch.Tensor):
    """(best_index, best_score) with the reference's strict-> / first-maximum rule."""
    if scores.numel() == 0:
        return -1, 0.0
    s = torch.nan_to_num(scores, nan=0.0)
    m = torch.max(s)
    if not bool(m > 0):
        return -1, 0.0
    idx = int(torch.nonzero(s == m)[0, 0])
    return idx, float(m)


def combine_scores(scores_all: torch.Tensor, group=None):
    """In-place all-reduce(SUM) of a vector in which this rank filled only its own slice."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(scores_all, op=dist.ReduceOp.SUM, group=group)
    return scores_all


class ShardedScorer:
    """Scores a global hypothesis batch across the ranks of the default process group.

    `score_local(T_slice) -> 1-D float32 tensor of len(T_slice)` is the per-rank scorer: on the GPU
    box a closure over LcpScorer.score_device, in the CPU tests any stand-in.
    `settle(T_all, scores_all) -> (best_index, best_score)` (optional) takes the arg-max over the
    combined vector on the device -- a closure over LcpScorer.settle_best_device, which also settles
    weighted near-ties ACROSS slices in the reference's summation order; without it the arg-max is
    best_of() over the combined values."""

    def __init__(self, score_local, rank=None, world=None, group=None, settle=None):
        self.score_local = score_local
        self.settle = settle
        self.group = group
        init = dist.is_available() and dist.is_initialized()
        self.rank = rank if rank is not None else (dist.get_rank(group) if init else 0)
        self.world = world if world is not None else (dist.get_world_size(group) if init else 1)

    def score(self, T_all: torch.Tensor):
        n_total = int(T_all.shape[0])
        lo, hi = shard_bounds(n_total, self.rank, self.world)
        out = torch.zeros(n_total, dtype=torch.float32, device=T_all.device)
        if hi > lo:
            out[lo:hi] = self.score_local(T_all[lo:hi])
        combine_scores(out, self.group)
        bi, bs = self.settle(T_all, out) if self.settle is not None else best_of(out)
        return out, bi, bs


class BucketedExchange:
    """Throughput form of the exchange for callers that stream MANY batches (bench.py, a search that
    scores the batches of several objects back to back): the score vectors of `bucket` consecutive
    batches share ONE all-reduce (a collective launch costs the scoring stream ~11 us, messages are
    4 B x n_total, far below the bandwidth regime), two buckets alternate so that the collective of
    one runs on the backend's stream while the scoring kernels fill the other, and the tail of an
    exchange (wait, arg-max, re-zeroing) runs on its own stream.  Every batch's vector is still
    all-reduced in full and arg-maxed; what the caller gives up is LATENCY -- a batch's combined
    scores exist up to `bucket` batches later.  bucket=1 is the per-call form (== ShardedScorer).

        ex = BucketedExchange(n_local, rank, world, device, bucket=8)
        for every batch:  scorer(ex.slot()); ex.commit()
        ex.drain();  ex.argmax  # arg-max per batch of the last completed bucket
    """

    def __init__(self, n_local, rank, world, device, bucket=8, group=None, force=False):
        self.n_local, self.rank, self.world = int(n_local), int(rank), int(world)
        self.device = torch.device(device)
        self.group = group
        self.cuda = self.device.type == "cuda"
        self.active = force or (dist.is_available() and dist.is_initialized() and world > 1)
        self.bucket = int(bucket) if self.active else 1
        n_buf = 2 if self.active else 1
        self.bufs = [torch.zeros(self.bucket, self.world * self.n_local, dtype=torch.float32, device=self.device)
                     for _ in range(n_buf)]
        self.works = [None] * n_buf
        self.k = 0
        self.open = None
        self.last = (0, 0)
        self.argmax = None
        if self.cuda and self.active:
            self.main = torch.cuda.current_stream(self.device)
            self.post = torch.cuda.Stream(self.device)
            self.ready = [torch.cuda.Event() for _ in range(n_buf)]

    def _finish(self, b):
        if self.works[b] is not None:
            self.works[b].wait()          # stream-level wait under nccl; host wait under gloo
            self.works[b] = None
            self.argmax = torch.argmax(self.bufs[b], dim=1)

    def _exchange(self, b):
        self.works[b] = dist.all_reduce(self.bufs[b], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.open = None

    def slot(self):
        """This rank's slice of the current batch's vector (score into it, then commit())."""
        if not self.active:
            self.last = (0, 0)
            return self.bufs[0][0, self.rank * self.n_local:(self.rank + 1) * self.n_local]
        b, j = (self.k // self.bucket) % len(self.bufs), self.k % self.bucket
        if j == 0:                       # a new bucket: its previous contents are consumed and cleared
            if self.cuda:
                with torch.cuda.stream(self.post):
                    self._finish(b)
                    self.bufs[b].zero_()  # every rank fills only its slice of a zeroed vector: sum == gather
                    self.ready[b].record(self.post)
                self.main.wait_event(self.ready[b])
            else:
                self._finish(b)
                self.bufs[b].zero_()
            self.open = b
        self.last = (b, j)
        return self.bufs[b][j, self.rank * self.n_local:(self.rank + 1) * self.n_local]

    def commit(self):
        if not self.active:
            return
        b, j = self.last
        self.k += 1
        if j == self.bucket - 1:
            self._exchange(b)

    def drain(self):
        """Exchange a partly filled bucket and complete everything in flight."""
        if not self.active:
            return
        if self.open is not None:
            self._exchange(self.open)
        # the bucket of the most recent batch last, so that `argmax` ends on it
        order = [b for b in range(len(self.bufs)) if b != self.last[0]] + [self.last[0]]
        if self.cuda:
            with torch.cuda.stream(self.post):
                for b in order:
                    self._finish(b)
            self.main.wait_stream(self.post)
        else:
            for b in order:
                self._finish(b)
        self.k = 0

    def last_vector(self):
        """The combined vector of the most recently committed batch (valid after drain())."""
        b, j = self.last
        return self.bufs[b][j]


class MultiObjectShardedScorer:
    """`score_local[obj](T_slice) -> scores` per object; every object's clouds are replicated on
    every rank.  One all-reduce for the concatenated score vector of all objects, then a local
    arg-max per object."""

    def __init__(self, score_local, rank=None, world=None, group=None):
        self.score_local = list(score_local)
        self.group = group
        init = dist.is_available() and dist.is_initialized()
        self.rank = rank if rank is not None else (dist.get_rank(group) if init else 0)
        self.world = world if world is not None else (dist.get_world_size(group) if init else 1)

    def score(self, T_per_object):
        counts = [int(T.shape[0]) for T in T_per_object]
        offs = [0]
        for c in counts:
            offs.append(offs[-1] + c)
        dev = T_per_object[0].device if counts else torch.device("cpu")
        flat = torch.zeros(offs[-1], dtype=torch.float32, device=dev)
        for obj, lo, hi in flat_slices(counts, self.rank, self.world):
            flat[offs[obj] + lo:offs[obj] + hi] = self.score_local[obj](T_per_object[obj][lo:hi])
        combine_scores(flat, self.group)
        per_obj = [flat[offs[o]:offs[o + 1]] for o in range(len(counts))]
        return per_obj, [best_of(s) for s in per_obj]
