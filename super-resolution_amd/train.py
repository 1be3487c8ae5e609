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

    The batch is split into ``n_streams`` sub-batches that run the same recorded op lists
    on separate HIP streams: at small patch sizes one convolution launch cannot fill 256
    CUs for its whole duration (prologue, epilogue, launch boundary), and two independent
    launch chains in flight hide each other's bubbles.  Sub-batch gradients accumulate
    into one flat buffer.  ``grad_sync`` (optional) is called with that buffer after
    backward and before the optimizer step (the data-parallel all-reduce hook)."""

    def __init__(self, gen, batch_shape, compute_dtype=torch.bfloat16, lr=2e-4, grad_sync=None,
                 n_streams: int = 1):
        B, _, h, w = batch_shape
        if B % n_streams:
            n_streams = 1
        self.gen = gen
        self.flat = flatten_parameters(gen)
        params = {n: p.data for n, p in gen.named_parameters()}
        dev = self.flat.device
        kw = dict(in_ch=gen.conv_first.in_channels, out_ch=gen.conv_last.out_channels,
                  nf=gen.conv_first.out_channels,
                  gc=gen.body[0].rdb1.conv1.out_channels if len(gen.body) else 32,
                  num_blocks=len(gen.body), need_grad=True, need_input_grad=False)
        self.n_streams = n_streams
        self.sub = B // n_streams
        self.plans = []
        for i in range(n_streams):
            fg = self.plans[0].flat_grad if i else None
            self.plans.append(engine.GeneratorPlan(params, self.sub, h, w, compute_dtype, flat_grad=fg,
                                                   accumulate_grads=(i > 0), **kw))
        self.plan = self.plans[0]
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)] if n_streams > 1 else [None]
        self.opt = FlatAdam(self.flat, lr=lr)
        self.loss_acc = torch.zeros(1, dtype=torch.float32, device=dev)
        self.grad_sync = grad_sync
        self.steps = 0

    def _sub_step(self, p, lr_img, hr_img):
        Y = p.forward_c8(lr_img)
        L.check(p.lib.srx_l1_c8_nchw(p.dt, Y.ptr(), Y.img_stride, hr_img.data_ptr(), self.loss_acc.data_ptr(),
                                     p.GY.ptr(), p.GY.img_stride, 1.0 / self.n_streams, Y.B, p.out_ch, Y.H, Y.W,
                                     engine.stream_ptr(p.device)), "srx_l1_c8_nchw")
        p.run_backward_ops()

    def step(self, lr_img: torch.Tensor, hr_img: torch.Tensor) -> None:
        hr_img = hr_img.contiguous()
        lr_img = lr_img.contiguous()
        if self.n_streams == 1:
            self._sub_step(self.plan, lr_img, hr_img)
            self.plan.run_unpack()
        else:
            main = torch.cuda.current_stream()
            ready = torch.cuda.Event()
            ready.record(main)
            done = []
            for i, (p, st) in enumerate(zip(self.plans, self.streams)):
                st.wait_event(ready)
                with torch.cuda.stream(st):
                    sl = slice(i * self.sub, (i + 1) * self.sub)
                    self._sub_step(p, lr_img[sl], hr_img[sl])
                    e = torch.cuda.Event()
                    e.record(st)
                    done.append(e)
            for e in done:
                main.wait_event(e)
            for p in self.plans:          # plan 0 writes, the others accumulate
                p.run_unpack()
        g = self.plan.flat_grad
        if self.grad_sync is not None:
            self.grad_sync(g)
        self.opt.step(g)
        self.steps += 1

    def mean_loss(self) -> float:
        """Average L1 loss since the last call (one host sync).  Sub-batch losses are each a
        mean over their own pixels, so their sum is divided by the stream count."""
        v = float(self.loss_acc.item()) / max(self.steps, 1) / self.n_streams
        self.loss_acc.zero_()
        self.steps = 0
        return v
