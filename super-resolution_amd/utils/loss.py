"""VGG-feature perceptual loss with the reference's interface (``utils/loss.py:8-22``),
executed by libsrx: VGG19 ``features[:36]`` of both images, L1 between the feature maps,
and the data gradient back to the first argument.

Weights: the reference downloads torchvision's ``IMAGENET1K_V1`` checkpoint.  There is no
network here, so the constructor looks for a local torchvision-format state_dict
(``SRX_VGG19_WEIGHTS`` or ``~/.cache/torch/hub/checkpoints/vgg19-dcbb9e9d.pth``) and raises
``FileNotFoundError`` if none is found: training against any other weights optimises a different
perceptual objective.  ``VGGLoss(allow_synthetic=True)`` (tests, benchmarks) falls back to seeded
synthetic weights instead and says so; ``weights_source`` records which were used.  Like the
reference, no ImageNet mean/std normalisation is applied to the inputs.
"""
from __future__ import annotations

import os
import warnings
from typing import Optional

import torch
from torch import nn

from .. import _lib as L
from .. import chain
from ..engine import stream_ptr
from ..host import _auto_compute_dtype

_HUB = os.path.expanduser("~/.cache/torch/hub/checkpoints/vgg19-dcbb9e9d.pth")


def build_vgg19_features36() -> nn.Sequential:
    """torchvision ``vgg19().features[:36]`` (cfg "E" without the last pool), indices 0..35."""
    layers, cin = [], 3
    for c in chain.VGG19_CFG:
        if c == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, c, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = c
    return nn.Sequential(*layers)


def _load_local_weights(vgg: nn.Sequential) -> Optional[str]:
    """Load a torchvision-format VGG19 state_dict if one is present; returns its path or None."""
    path = os.environ.get("SRX_VGG19_WEIGHTS") or (_HUB if os.path.exists(_HUB) else None)
    if not path:
        return None
    sd = torch.load(path, map_location="cpu", weights_only=True)
    sd = {k[len("features."):]: v for k, v in sd.items() if k.startswith("features.")} or sd
    vgg.load_state_dict({k: v for k, v in sd.items() if k.split(".")[0].isdigit() and int(k.split(".")[0]) < 36})
    return path


class _VGGLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fake, real, anchor, host):
        loss, plan = host._forward_impl(fake, real, need_grad=fake.requires_grad)
        ctx.host, ctx.plan, ctx.version = host, plan, plan.version
        ctx.in_dtype = fake.dtype
        return loss

    @staticmethod
    def backward(ctx, gloss):
        return ctx.host._backward_impl(ctx.plan, ctx.version, gloss).to(ctx.in_dtype), None, None, None


class VGGLoss(nn.Module):
    """
    Calculates the L1 loss between the feature maps of a pre-trained VGG19 network.
    This is used as the Perceptual Loss component.
    """

    compute_dtype: Optional[torch.dtype] = None

    def __init__(self, device=None, seed: int = 0, allow_synthetic: bool = False):
        super().__init__()
        vgg = build_vgg19_features36().eval()
        path = _load_local_weights(vgg)
        self.weights_source = path or "synthetic (seed %d)" % seed
        if path is None:
            if not allow_synthetic:
                raise FileNotFoundError(
                    "VGGLoss: no VGG19 ImageNet checkpoint found.  Point SRX_VGG19_WEIGHTS at a torchvision "
                    "vgg19-dcbb9e9d.pth state_dict (or place it at %s).  VGGLoss(allow_synthetic=True) uses "
                    "seeded random weights instead: fine for tests and benchmarks, not for training." % _HUB)
            warnings.warn("VGGLoss: no local VGG19 checkpoint found (set SRX_VGG19_WEIGHTS); using seeded "
                          "synthetic weights", stacklevel=2)
            g = torch.Generator().manual_seed(seed)
            with torch.no_grad():
                for m in vgg:
                    if isinstance(m, nn.Conv2d):
                        fan_in = m.in_channels * 9
                        m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5)
                        m.bias.zero_()
        for param in vgg.parameters():
            param.requires_grad = False
        if device is None:
            device = "cuda" if torch.cuda.is_available() else "cpu"
        self.vgg = vgg.to(device)
        self.loss = nn.L1Loss()

    def _apply(self, fn, *a, **k):
        self.__dict__.pop("_plans", None)
        return super()._apply(fn, *a, **k)

    def _get_plans(self, x: torch.Tensor, need_grad: bool):
        tdtype = _auto_compute_dtype(self, x)
        key = (tuple(x.shape), tdtype, need_grad)
        plans = self.__dict__.setdefault("_plans", {})
        if key not in plans:
            params = {n: p.data.float() if p.dtype != torch.float32 else p.data
                      for n, p in self.vgg.named_parameters()}
            B, _, H, W = x.shape
            pf = chain.VGGFeaturePlan(params, B, H, W, tdtype, need_input_grad=need_grad)
            pr = chain.VGGFeaturePlan(params, B, H, W, tdtype, need_input_grad=False)
            loss_buf = torch.zeros(1, dtype=torch.float32, device=x.device)
            pf.version = 0
            plans[key] = (pf, pr, loss_buf, params)
            pf.run_packing()
            pr.run_packing()          # frozen weights: packed once
        return plans[key]

    def _forward_impl(self, fake, real, need_grad: bool):
        if fake.device.type != "cuda":
            raise RuntimeError("VGGLoss runs on a ROCm GPU only: the HIP extension is the only implementation")
        pf, pr, loss_buf, _ = self._get_plans(fake, need_grad)
        pf.version += 1
        fa = pf.forward_c8(fake, pack=False)
        fb = pr.forward_c8(real.detach(), pack=False)
        loss_buf.zero_()
        g = pf.gA[-1] if need_grad else None
        B, C_, H, W = pf.geo[-1]
        L.check(pf.lib.srx_l1_c8(pf.dt, fa.ptr(), fa.img_stride, fb.ptr(), fb.img_stride, loss_buf.data_ptr(),
                                 g.ptr() if g is not None else None, g.img_stride if g is not None else 0,
                                 1.0, 0.0, B, C_, H, W, stream_ptr(pf.device)), "srx_l1_c8")
        return loss_buf[0].clone(), pf

    def _backward_impl(self, pf, version: int, gloss: torch.Tensor) -> torch.Tensor:
        if pf.version != version:
            raise RuntimeError("VGGLoss: the activations saved for this backward were overwritten by a later "
                               "forward of the same shape (one un-backpropagated forward per shape at a time)")
        pf.backward_c8()
        return pf.c8_to_nchw((pf.gA[0], 0), 3) * gloss

    def forward(self, input_features, target_features):
        if torch.is_grad_enabled() and input_features.requires_grad:
            anchor = torch.empty(0, device=input_features.device, requires_grad=True)
            return _VGGLossFn.apply(input_features, target_features, anchor, self)
        return self._forward_impl(input_features, target_features, need_grad=False)[0]
