"""Flat parameter storage: every parameter of a module becomes a view of one fp32
buffer (in ``parameters()`` order), so the optimizer step and the data-parallel
gradient exchange are single launches / a few large collectives over contiguous
memory instead of one per tensor."""
from __future__ import annotations

from typing import List, Tuple

import torch
from torch import nn


def flatten_parameters(module: nn.Module) -> torch.Tensor:
    """Re-point ``p.data`` of every parameter at a slice of one new flat tensor
    (values preserved).  Returns the flat tensor.  Optimizers created earlier keep
    working: they hold the Parameter objects, not their storage."""
    params: List[nn.Parameter] = list(module.parameters())
    if not params:
        raise ValueError("module has no parameters")
    dev, dt = params[0].device, params[0].dtype
    total = sum(p.numel() for p in params)
    flat = torch.empty(total, dtype=dt, device=dev)
    off = 0
    with torch.no_grad():
        for p in params:
            n = p.numel()
            flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = flat[off:off + n].view(p.shape)
            off += n
    return flat


def param_offsets(module: nn.Module) -> List[Tuple[str, int, int]]:
    out, off = [], 0
    for n, p in module.named_parameters():
        out.append((n, off, p.numel()))
        off += p.numel()
    return out
