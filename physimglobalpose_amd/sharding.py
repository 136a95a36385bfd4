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
    box a closure over LcpScorer.score_device, in the CPU tests any stand-in."""

    def __init__(self, score_local, rank=None, world=None, group=None):
        self.score_local = score_local
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
        bi, bs = best_of(out)
        return out, bi, bs


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
