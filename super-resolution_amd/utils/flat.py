"""Flat parameter storage: every parameter of a module becomes a view of one fp32
buffer (in ``parameters()`` order), so the optimizer step and the data-parallel
gradient exchange are single launches / a few large collectives over contiguous
memory instead of one per tensor."""
from __future__ import annotations

from typing import List, Tuple

import torch
from torch import nn


def flatten_parameters(module: nn.Module) -> torch.Tensor:
    """Re-point ``p.data`` of every parameter at a slice of one flat tensor (values
    preserved) and return it; a second call on an intact module returns the same tensor.
    Optimizers created earlier keep working: they hold the Parameter objects, not their
    storage."""
    params: List[nn.Parameter] = list(module.parameters())
    if not params:
        raise ValueError("module has no parameters")
    dev, dt = params[0].device, params[0].dtype
    total = sum(p.numel() for p in params)
    # Already flat (an earlier trainer of the same module did it) and still intact — .to() or
    # .half() since then would have given the parameters new storage: hand out the same buffer,
    # so that every trainer, optimizer and plan of the module keeps seeing one set of weights.
    prev = module.__dict__.get("_srx_flat")
    if prev is not None and prev.numel() == total and prev.device == dev and prev.dtype == dt:
        off, ok = 0, True
        for p in params:
            ok = ok and p.data.data_ptr() == prev.data_ptr() + off * prev.element_size() and p.data.is_contiguous()
            off += p.numel()
        if ok:
            return prev
    flat = torch.empty(total, dtype=dt, device=dev)
    off = 0
    with torch.no_grad():
        for p in params:
            n = p.numel()
            flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = flat[off:off + n].view(p.shape)
            off += n
    module.__dict__["_srx_flat"] = flat
    return flat


def param_offsets(module: nn.Module) -> List[Tuple[str, int, int]]:
    out, off = [], 0
    for n, p in module.named_parameters():
        out.append((n, off, p.numel()))
        off += p.numel()
    return out
