"""Data-parallel gradient exchange: one process per GPU, torch.distributed over RCCL
(backend "nccl" on ROCm) or gloo on CPU for tests.

The reference has no distributed code; batch sharding with a gradient average is the
only exchange the path needs (no BatchNorm or other cross-sample op exists in G, D or
VGG; spectral-norm u/v evolve identically on every rank and are never communicated).
Gradients live in ONE flat fp32 buffer per model, so the exchange is a few large
bucketed all-reduces rather than 702 small ones.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


def _host_staged_sum(t: torch.Tensor, group) -> None:
    """Sum a device tensor over a gloo group through a host copy.  Only rehearsals of the
    multi-rank path on a single GPU come here (RCCL needs one GPU per rank); gloo's own
    device-tensor collectives stall intermittently when two ranks share one GPU."""
    # every kernel that produces t has finished before the copy is queued: with three processes on
    # the card (two ranks and the test runner) a device-to-host copy queued behind cross-stream
    # event waits was seen to stall for tens of seconds
    torch.cuda.synchronize(t.device)
    h = t.detach().cpu()
    dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
    t.copy_(h)


def _multi_rank(group=None) -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def broadcast_from_rank0(tensors, group=None) -> None:
    """Make every rank's copy of the given tensors (flat parameter buffers, spectral-norm u / v)
    equal to rank 0's.  Data-parallel training is only correct while the replicas hold the same
    parameters; building the models from the same seed on every rank gives that too, and this
    makes it hold by construction.  No-op without a process group or with one rank."""
    if not _multi_rank(group):
        return
    staged = dist.get_backend(group) != "nccl"
    for t in tensors:
        if t.is_cuda and staged:                    # gloo rehearsal with device tensors
            h = t.detach().cpu()
            dist.broadcast(h, src=0, group=group)
            t.copy_(h)
        else:
            dist.broadcast(t, src=0, group=group)


def barrier(group=None) -> None:
    """Every rank waits here (a file rank 0 has just written is complete before any rank reads
    it).  No-op without a process group or with one rank."""
    if _multi_rank(group):
        dist.barrier(group=group)


class FlatGradAllReduce:
    """Average a flat gradient buffer over the ranks in buckets.

    Buckets are issued back to back as async collectives on the process group's own
    stream and waited for together, so RCCL pipelines them; bucket size is chosen for
    xGMI's per-link-bound rings (a few tens of MB keeps every link busy without
    serialising the whole 68 MB generator gradient behind one launch)."""

    def __init__(self, world_size: Optional[int] = None, bucket_mb: float = 24.0, group=None,
                 single_rank_too: bool = False):
        self.group = group
        self.world = world_size if world_size is not None else dist.get_world_size(group)
        # a one-rank group normally skips the collectives; single_rank_too issues them anyway
        # (a rehearsal of the whole exchange on one GPU)
        self.active = self.world > 1 or single_rank_too
        self.bucket_elems = max(1, int(bucket_mb * (1 << 20) // 4))

    def buckets(self, n: int) -> List[slice]:
        return [slice(i, min(n, i + self.bucket_elems)) for i in range(0, n, self.bucket_elems)]

    def __call__(self, flat_grad: torch.Tensor) -> None:
        if not self.active:
            return
        use_avg = dist.get_backend(self.group) == "nccl"        # gloo has no AVG
        op = dist.ReduceOp.AVG if use_avg else dist.ReduceOp.SUM
        works = []
        for s in self.buckets(flat_grad.numel()):
            if flat_grad.is_cuda and not use_avg:
                _host_staged_sum(flat_grad[s], self.group)
            else:
                works.append(dist.all_reduce(flat_grad[s], op=op, group=self.group, async_op=True))
        for w in works:
            w.wait()
        if not use_avg:
            flat_grad.div_(self.world)


class OverlappedGradAllReduce(FlatGradAllReduce):
    """The same average, issued piecewise while backward is still running.

    ``begin(flat)`` once per step; ``ready(lo, hi)`` whenever ``flat[lo:hi]`` is final — called
    with the stream that produced it current, so the collective (which runs on the process
    group's stream behind an event on the current stream) starts as soon as that range is
    written and overlaps everything backward still has to do; ``finish()`` before the
    optimizer step.  Adjacent ranges are merged until a bucket is full: backward completes the
    generator's gradients from the last layer to the first, so buckets fill in reverse
    parameter order."""

    def begin(self, flat_grad: torch.Tensor) -> None:
        self._flat = flat_grad
        self._works = []
        self._pending: Optional[List[int]] = None
        backend = dist.get_backend(self.group) if self.active else ""
        self._avg = backend == "nccl"

    def _flush(self) -> None:
        if self._pending is None:
            return
        lo, hi = self._pending
        self._pending = None
        if self.active:
            op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
            if self._flat.is_cuda and not self._avg:
                _host_staged_sum(self._flat[lo:hi], self.group)
            else:
                self._works.append(dist.all_reduce(self._flat[lo:hi], op=op, group=self.group, async_op=True))

    def ready(self, lo: int, hi: int) -> None:
        if self._pending is not None and hi == self._pending[0]:
            self._pending[0] = lo
        elif self._pending is not None and lo == self._pending[1]:
            self._pending[1] = hi
        else:
            self._flush()
            self._pending = [lo, hi]
        if self._pending[1] - self._pending[0] >= self.bucket_elems:
            self._flush()

    def finish(self) -> None:
        self._flush()
        for w in self._works:
            w.wait()
        self._works = []
        if self.active and not self._avg:
            self._flat.div_(self.world)


def shard_batch(x: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """The rank's contiguous slice of a global batch (global batch must divide evenly)."""
    if x.shape[0] % world:
        raise ValueError(f"global batch {x.shape[0]} is not divisible by world size {world}")
    per = x.shape[0] // world
    return x[rank * per:(rank + 1) * per]
