"""Validation pass of the ESRGAN trainer (reference ``esrgan/train_esrgan.py:43-77``):
per batch, PSNR of the generator output and of a bicubic x4 upscale against the HR target,
with the reference's convention PSNR = 10*log10(1/MSE) on [-1,1] data (peak taken as 1).

The generator runs through its inference plan and the squared error is reduced by
``srx_sqerr_c8_nchw`` straight from the generator's C8 output; the bicubic baseline is a
plain ``F.interpolate`` (it is a yardstick, not part of the model path).  Per-batch values
stay on the device: one host synchronisation per validation pass instead of two per batch.
"""
from __future__ import annotations

from typing import Dict, Iterable, Tuple

import torch
import torch.nn.functional as F

from . import _lib as L
from . import engine
from .host import ParamTracker

SCALE = 4


class Validator:
    def __init__(self, gen, compute_dtype=torch.bfloat16):
        self.gen = gen
        self.tdtype = compute_dtype
        self._plans: Dict[Tuple[int, int, int], engine.GeneratorPlan] = {}
        self._tracker = ParamTracker()

    def _plan(self, B, h, w):
        g = self.gen
        # a trainer built after this object (flatten_parameters), gen.to() or gen.half() move the
        # parameters to new storage: plans that point at the old one are dropped
        self._params, moved = self._tracker.current(g)
        if moved:
            self._plans.clear()
        key = (B, h, w)
        if key not in self._plans:
            self._plans[key] = engine.GeneratorPlan(
                self._params, B, h, w, self.tdtype, in_ch=g.conv_first.in_channels,
                out_ch=g.conv_last.out_channels, nf=g.conv_first.out_channels,
                gc=g.body[0].rdb1.conv1.out_channels if len(g.body) else 32, num_blocks=len(g.body),
                need_grad=False, need_input_grad=False)
        return self._plans[key]

    @torch.no_grad()
    def batch_mse(self, lr: torch.Tensor, hr: torch.Tensor, out: torch.Tensor) -> None:
        """out[0] += sum over the batch of (G(lr) - hr)^2; the caller divides by hr.numel()."""
        B, _, h, w = lr.shape
        plan = self._plan(B, h, w)
        Y = plan.forward_c8(lr.contiguous().float())
        hr = hr.contiguous().float()
        L.check(plan.lib.srx_sqerr_c8_nchw(plan.dt, Y.ptr(), Y.img_stride, hr.data_ptr(), out.data_ptr(),
                                           B, plan.out_ch, Y.H, Y.W, engine.stream_ptr(plan.device)),
                "srx_sqerr_c8_nchw")
        self._hold = hr

    @torch.no_grad()
    def validate_one_epoch(self, loader: Iterable) -> Dict[str, float]:
        """Returns {'val_sr_psnr', 'val_bicubic_psnr', 'improvement'} — the three numbers the
        reference prints (:73-76); 'val_sr_psnr' is what it returns."""
        was_training = self.gen.training
        self.gen.eval()
        dev = next(self.gen.parameters()).device
        sr, bi = [], []
        for lr, hr in loader:
            lr, hr = lr.to(dev), hr.to(dev)
            acc = torch.zeros(1, dtype=torch.float32, device=dev)
            self.batch_mse(lr, hr, acc)
            sr.append(10 * torch.log10(hr.numel() / acc[0]))                     # 10*log10(1/mse)
            up = F.interpolate(lr.float(), scale_factor=SCALE, mode="bicubic", align_corners=False)
            bi.append(10 * torch.log10(1 / F.mse_loss(up, hr.float())))
        self.gen.train(was_training)
        if not sr:
            raise ValueError("empty validation loader")
        avg_sr = float(torch.stack(sr).mean())
        avg_bi = float(torch.stack(bi).mean())
        return {"val_sr_psnr": avg_sr, "val_bicubic_psnr": avg_bi,
                "improvement": 100 * (1 - avg_sr / avg_bi)}
