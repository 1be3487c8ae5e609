"""Native training harness for the ESRGAN hot path (counterpart of the reference's
``esrgan/train_esrgan.py:80-118`` step and of its stubbed warm-up loop, :166-171).

Everything between "batch is on the device" and "parameters are updated" runs in
libsrx: generator forward, fused L1 loss + gradient, backward, (data-parallel
gradient exchange), fused Adam on flat parameter storage.  No host
synchronisation happens inside a step; losses accumulate in a device buffer that
the caller reads whenever it wants to log.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib as L
from . import engine
from .utils.flat import flatten_parameters


class FlatAdam:
    """torch.optim.Adam(lr, betas=(0.9,0.999)) semantics (esrgan/train_esrgan.py:134-135)
    over flat fp32 parameter / gradient buffers: one srx_adam launch per step."""

    def __init__(self, flat_param: torch.Tensor, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.p = flat_param
        self.m = torch.zeros_like(flat_param)
        self.v = torch.zeros_like(flat_param)
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.step_count = 0

    def step(self, flat_grad: torch.Tensor, grad_scale: float = 1.0):
        self.step_count += 1
        L.check(L.lib().srx_adam(self.p.data_ptr(), flat_grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                 self.p.numel(), self.lr, self.betas[0], self.betas[1], self.eps, self.wd,
                                 self.step_count, grad_scale, engine.stream_ptr(self.p.device)), "srx_adam")


class GeneratorWarmupTrainer:
    """Generator warm-up: L1 pixel loss, fwd + bwd + Adam (BASELINE config 2).

    ``grad_sync`` (optional) is called with the flat gradient buffer after backward
    and before the optimizer step (the data-parallel all-reduce hook)."""

    def __init__(self, gen, batch_shape, compute_dtype=torch.bfloat16, lr=2e-4, grad_sync=None):
        B, _, h, w = batch_shape
        self.gen = gen
        self.flat = flatten_parameters(gen)
        params = {n: p.data for n, p in gen.named_parameters()}
        self.plan = engine.GeneratorPlan(
            params, B, h, w, compute_dtype, in_ch=gen.conv_first.in_channels,
            out_ch=gen.conv_last.out_channels, nf=gen.conv_first.out_channels,
            gc=gen.body[0].rdb1.conv1.out_channels if len(gen.body) else 32, num_blocks=len(gen.body),
            need_grad=True, need_input_grad=False)
        self.opt = FlatAdam(self.flat, lr=lr)
        self.loss_acc = torch.zeros(1, dtype=torch.float32, device=self.flat.device)
        self.grad_sync = grad_sync
        self.steps = 0

    def step(self, lr_img: torch.Tensor, hr_img: torch.Tensor) -> None:
        p = self.plan
        Y = p.forward_c8(lr_img)
        hr_img = hr_img.contiguous()
        L.check(p.lib.srx_l1_c8_nchw(p.dt, Y.ptr(), Y.img_stride, hr_img.data_ptr(), self.loss_acc.data_ptr(),
                                     p.GY.ptr(), p.GY.img_stride, 1.0, Y.B, p.out_ch, Y.H, Y.W,
                                     engine.stream_ptr(p.device)), "srx_l1_c8_nchw")
        p.backward_c8()
        if self.grad_sync is not None:
            self.grad_sync(p.flat_grad)
        self.opt.step(p.flat_grad)
        self.steps += 1

    def mean_loss(self) -> float:
        """Average L1 loss since the last call (one host sync)."""
        v = float(self.loss_acc.item()) / max(self.steps, 1)
        self.loss_acc.zero_()
        self.steps = 0
        return v
