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


class FlatGradAllReduce:
    """Average a flat gradient buffer over the ranks in buckets.

    Buckets are issued back to back as async collectives on the process group's own
    stream and waited for together, so RCCL pipelines them; bucket size is chosen for
    xGMI's per-link-bound rings (a few tens of MB keeps every link busy without
    serialising the whole 68 MB generator gradient behind one launch)."""

    def __init__(self, world_size: Optional[int] = None, bucket_mb: float = 24.0, group=None):
        self.group = group
        self.world = world_size if world_size is not None else dist.get_world_size(group)
        self.bucket_elems = max(1, int(bucket_mb * (1 << 20) // 4))

    def buckets(self, n: int) -> List[slice]:
        return [slice(i, min(n, i + self.bucket_elems)) for i in range(0, n, self.bucket_elems)]

    def __call__(self, flat_grad: torch.Tensor) -> None:
        if self.world == 1:
            return
        avg = dist.ReduceOp.AVG if flat_grad.is_cuda else dist.ReduceOp.SUM
        works = [dist.all_reduce(flat_grad[s], op=avg, group=self.group, async_op=True)
                 for s in self.buckets(flat_grad.numel())]
        for w in works:
            w.wait()
        if not flat_grad.is_cuda:
            flat_grad.div_(self.world)


def shard_batch(x: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """The rank's contiguous slice of a global batch (global batch must divide evenly)."""
    if x.shape[0] % world:
        raise ValueError(f"global batch {x.shape[0]} is not divisible by world size {world}")
    per = x.shape[0] // world
    return x[rank * per:(rank + 1) * per]
