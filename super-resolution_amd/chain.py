"""Plans for the two feed-forward convolution chains of the ESRGAN step: the
spectral-normalised discriminator (esrgan/discriminator.py:5-38) and the VGG19
feature extractor of the perceptual loss (utils/loss.py:8-22).

A chain is a list of layer specs executed in order:
  conv   3x3 (stride 1/2) or 1x1 + optional LeakyReLU/ReLU  -> srx_conv2d
  pool   MaxPool2d(2,2)                                      -> srx_maxpool2_*
  gap    AdaptiveAvgPool2d(1); its output is laid out as ONE image whose width is
         the batch, so the 1x1 classifier convolutions see M = B pixels.
Backward walks the chain in reverse.  Every backward op multiplies the gradient it
produces by the activation derivative of the tensor it flows into (taken from the
sign of the saved forward output), so what reaches a convolution is already the
gradient w.r.t. its pre-activation output: that is what its weight gradient and
its data gradient consume.  The data gradient of a stride-2 convolution is run as
a stride-1 convolution over the zero-stuffed gradient.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import os

import torch

from . import _lib as L
from .engine import Buf, OpList, Plan, View, _ceil, conv_seg, dgrad_seg, stream_ptr


class _GradMixin:
    """Flat gradient storage (parameter order), tap-major staging and the unpack table."""

    def setup_grads(self, P: Dict[str, torch.Tensor], names: List[str], flat_grad=None):
        self.param_names = list(names)
        total = sum(P[n].numel() for n in names)
        self.flat_grad = flat_grad if flat_grad is not None else self.f32(total, zero=True)
        self.grads: Dict[str, torch.Tensor] = {}
        self._stage_total = total
        self.stage = self.f32(total * (self.det_slices if self.deterministic_wgrad else 1), zero=True)
        self._soff: Dict[str, int] = {}
        off = 0
        for n in names:
            t = P[n]
            self.grads[n] = self.flat_grad[off:off + t.numel()].view(t.shape)
            self._soff[n] = off
            if t.ndim == 4:
                self.unpack_item(self.sp(n), self.grads[n].data_ptr(), t.shape[0], t.shape[1],
                                 t.shape[2] * t.shape[3])
            else:
                self.unpack_item(self.sp(n), self.grads[n].data_ptr(), 1, t.numel(), 1)
            off += t.numel()

    def sp(self, name: str) -> int:
        return self.stage.data_ptr() + 4 * self._soff[name]


class ChainPlan(Plan, _GradMixin):
    """layers: list of dicts
        {"kind": "conv", "w": Tensor[cout,cin,k,k], "b": Tensor[cout] | None, "wname": str, "bname": str,
         "stride": 1|2, "act": slope (1.0 = none), "scale_ptr": device float* (0 = none)}
        {"kind": "pool"} | {"kind": "gap"}
    """

    # workgroups (tasks x pixel slices) a layer's weight-gradient launch aims for
    wgrad_blocks = int(os.environ.get("SRX_CHAIN_WGRAD_BLOCKS", "512"))

    def __init__(self, layers: List[dict], B: int, C_in: int, H: int, W: int, tdtype, device, *,
                 need_param_grad: bool, need_input_grad: bool, final_slope: float = 1.0,
                 accumulate_grads: bool = False):
        super().__init__(device, tdtype)
        self.unpack_accumulate = 1 if accumulate_grads else 0
        self.layers = layers
        self.B, self.C_in, self.H, self.W = B, C_in, H, W
        self.need_param_grad, self.need_input_grad = need_param_grad, need_input_grad
        need_bwd = need_param_grad or need_input_grad
        fwd, bwd = OpList(), OpList()

        # ---------------------------------------------------------------- forward
        self.A: List[Buf] = [self.buf(B, C_in, H, W)]
        self.slopes: List[float] = [1.0]          # activation slope that produced A[i] (1 = none)
        geo = [(B, C_in, H, W)]                    # (batch, channels, h, w) of A[i]
        b_, c_, h_, w_ = B, C_in, H, W
        fwd_pack = []
        for ly in layers:
            k = ly["kind"]
            if k == "conv":
                wt = ly["w"]
                cout, cin, ks = wt.shape[0], wt.shape[1], wt.shape[2]
                assert cin == c_, f"layer expects {cin} channels, chain has {c_}"
                st = ly.get("stride", 1)
                oh, ow = ((h_ + 2 * (ks // 2) - ks) // st + 1, (w_ + 2 * (ks // 2) - ks) // st + 1)
                out = self.buf(b_, cout, oh, ow)
                wp, bp = self.packed_layer(cout, cin, ks * ks, 0,
                                           [conv_seg(wt, 1.0, ly.get("scale_ptr", 0))],
                                           ly["b"].data_ptr() if ly.get("b") is not None else 0)
                self.conv(fwd, (self.A[-1], 0), (out, 0), wp, bp, b_, h_, w_, cin, cout, ksize=ks,
                          stride=st, act=ly.get("act", 1.0))
                fwd_pack.append((wp, bp))
                c_, h_, w_ = cout, oh, ow
                self.slopes.append(ly.get("act", 1.0))
            elif k == "pool":
                out = self.buf(b_, c_, h_ // 2, w_ // 2)
                self.generic(fwd, L.OP_MAXPOOL_FWD, a=(self.A[-1], 0), c=(out, 0), B=b_, H=h_, W=w_,
                             noct=_ceil(c_, 8))
                h_, w_ = h_ // 2, w_ // 2
                self.slopes.append(1.0)       # the pool's own backward applies its input's ReLU mask
                fwd_pack.append(None)
            elif k == "gap":
                out = self.buf(1, c_, 1, b_)
                self.generic(fwd, L.OP_GAP_FWD, a=(self.A[-1], 0), c=(out, 0), B=b_, H=h_, W=w_,
                             noct=_ceil(c_, 8))
                h_, w_, b_ = 1, b_, 1
                self.slopes.append(1.0)
                fwd_pack.append(None)
            else:
                raise ValueError(k)
            self.A.append(out)
            geo.append((b_, c_, h_, w_))
        self.geo = geo
        self.out_buf = self.A[-1]
        self.fwd_ops, self.fwd_meta = self.freeze(fwd), fwd.meta

        # --------------------------------------------------------------- backward
        self.bwd_ops, self.bwd_meta = None, []
        if need_bwd:
            P = {}
            names = []
            for ly in layers:
                if ly["kind"] == "conv" and need_param_grad:
                    P[ly["wname"]] = ly["w"]
                    names.append(ly["wname"])
                    if ly.get("b") is not None:
                        P[ly["bname"]] = ly["b"]
                        names.append(ly["bname"])
            if need_param_grad:
                # parameter order of the module (bias before weight for spectral-norm layers) is
                # restored by the caller through param_order
                order = self.param_order(names)
                self.setup_grads(P, order)
                self.memset(bwd, self.stage)
            n = len(layers)
            self.gA: List[Optional[Buf]] = [None] * (n + 1)
            self.gA[n] = self.buf(*[geo[n][0], geo[n][1], geo[n][2], geo[n][3]])
            for i in range(n, 0, -1):
                ly = layers[i - 1]
                b_i, c_i, h_i, w_i = geo[i - 1]          # geometry of the layer's INPUT
                produce_input = (i - 1 > 0) or need_input_grad
                if produce_input:
                    self.gA[i - 1] = self.buf(b_i, c_i, h_i, w_i)
                dst_mask = (self.A[i - 1], 0) if self.slopes[i - 1] != 1.0 else None
                dst_slope = self.slopes[i - 1]
                if ly["kind"] == "conv":
                    wt = ly["w"]
                    cout, cin, ks = wt.shape[0], wt.shape[1], wt.shape[2]
                    st = ly.get("stride", 1)
                    if need_param_grad:
                        tasks = self.wgrad_tasks_for_layer(
                            cin, cout, 0, 0, self.sp(ly["wname"]),
                            bias_ptr=self.sp(ly["bname"]) if ly.get("b") is not None else 0)
                        self.wgrad(bwd, (self.A[i - 1], 0), (self.gA[i], 0), tasks, b_i, h_i, w_i,
                                   ksize=ks, stride=st, target_blocks=self.wgrad_blocks)
                    if produce_input:
                        wp = self.packed_layer(cin, cout, ks * ks, 0,
                                               [dgrad_seg(wt, 0, 0, 1.0, ly.get("scale_ptr", 0))],
                                               with_bias=False)[0]
                        src: View = (self.gA[i], 0)
                        if st == 2:
                            zs = self.buf(b_i, cout, h_i, w_i)
                            oh, ow = geo[i][2], geo[i][3]
                            assert (2 * oh, 2 * ow) == (h_i, w_i), "stride-2 layers need even sizes"
                            self.generic(bwd, L.OP_ZERO_STUFF, a=(self.gA[i], 0), c=(zs, 0), B=b_i, H=oh, W=ow,
                                         noct=_ceil(cout, 8))
                            src = (zs, 0)
                        self.conv(bwd, src, (self.gA[i - 1], 0), wp, 0, b_i, h_i, w_i, cout, cin, ksize=ks,
                                  mask=dst_mask, mask_slope=dst_slope)
                elif ly["kind"] == "pool":
                    if produce_input:
                        self.generic(bwd, L.OP_MAXPOOL_BWD, a=(self.A[i - 1], 0), b=(self.gA[i], 0),
                                     c=(self.gA[i - 1], 0), mask=dst_mask, f0=dst_slope, B=b_i, H=h_i, W=w_i,
                                     noct=_ceil(c_i, 8))
                elif ly["kind"] == "gap":
                    if produce_input:
                        self.generic(bwd, L.OP_GAP_BWD, a=(self.gA[i], 0), c=(self.gA[i - 1], 0), mask=dst_mask,
                                     f0=dst_slope, B=b_i, H=h_i, W=w_i, noct=_ceil(c_i, 8))
            self.trim_stage_memset(bwd)
            self.bwd_ops, self.bwd_meta = self.freeze(bwd), bwd.meta
        self.finalize_packing()
        self.finalize_unpack()

    def param_order(self, names: List[str]) -> List[str]:
        return names

    # ------------------------------------------------------------------ run
    def forward_c8(self, x_nchw: Optional[torch.Tensor] = None, pack: bool = True) -> Buf:
        if pack:
            self.run_packing()
        if x_nchw is not None:
            self.nchw_to_c8(x_nchw, (self.A[0], 0))
        self.run(self.fwd_ops)
        return self.out_buf

    def backward_c8(self) -> None:
        """self.gA[-1] must hold the gradient w.r.t. the chain output (already multiplied by the
        final activation's derivative if there is one)."""
        self.run(self.bwd_ops)
        if self.need_param_grad:
            self.run_unpack()


# ===========================================================================
# Discriminator
# ===========================================================================
D_STRIDES = [1, 2, 1, 2, 1, 2, 1, 2]                      # esrgan/discriminator.py:18-25
D_LAYER_NAMES = [f"features.{i}.0" for i in range(8)] + ["classifier.1", "classifier.3"]


class DiscriminatorPlan(ChainPlan):
    """One forward (+ backward) of the spectral-normalised discriminator.

    params: name -> fp32 tensors (``*.bias``, ``*.weight_orig``); buffers: ``*.weight_u`` /
    ``*.weight_v`` (updated IN PLACE by the power iteration in training mode, like the
    reference hook does: torch spectral_norm.py:103-106).  The normalised weight is never
    materialised in fp32: the packer multiplies by 1/sigma read from device memory.
    """

    def __init__(self, params: Dict[str, torch.Tensor], buffers: Dict[str, torch.Tensor], B, H, W, tdtype,
                 *, need_param_grad=True, need_input_grad=True, accumulate_grads=False):
        dev = params["features.0.0.weight_orig"].device
        self.sn_names = D_LAYER_NAMES
        nl = len(self.sn_names)
        # private copies of u, v (the ones this forward used) + 1/sigma + scratch
        self._u = [torch.empty_like(buffers[n + ".weight_u"]) for n in self.sn_names]
        self._v = [torch.empty_like(buffers[n + ".weight_v"]) for n in self.sn_names]
        self._mod_u = [buffers[n + ".weight_u"] for n in self.sn_names]
        self._mod_v = [buffers[n + ".weight_v"] for n in self.sn_names]
        self.inv_sigma = torch.ones(nl, dtype=torch.float32, device=dev)
        self._wv = [torch.empty(params[n + ".weight_orig"].shape[0], dtype=torch.float32, device=dev)
                    for n in self.sn_names]
        self._dot = torch.zeros(nl, dtype=torch.float32, device=dev)
        layers = []
        for i, n in enumerate(self.sn_names):
            w = params[n + ".weight_orig"]
            layers.append({"kind": "conv", "w": w, "b": params[n + ".bias"], "wname": n + ".weight_orig",
                           "bname": n + ".bias", "stride": D_STRIDES[i] if i < 8 else 1,
                           "act": 0.2 if i != 9 else 1.0,
                           "scale_ptr": self.inv_sigma.data_ptr() + 4 * i})
            if i == 7:
                layers.append({"kind": "gap"})
        in_ch = params["features.0.0.weight_orig"].shape[1]
        super().__init__(layers, B, in_ch, H, W, tdtype, dev, need_param_grad=need_param_grad,
                         need_input_grad=need_input_grad, accumulate_grads=accumulate_grads)
        self.params = params
        # spectral-norm tables (device)
        sn = (L.SnLayer * nl)()
        snb = (L.SnBwdLayer * nl)()
        self.max_rows = self.max_cols = 0
        for i, n in enumerate(self.sn_names):
            w = params[n + ".weight_orig"]
            rows, cols = w.shape[0], w[0].numel()
            self.max_rows, self.max_cols = max(self.max_rows, rows), max(self.max_cols, cols)
            s = sn[i]
            s.w, s.u, s.v = w.data_ptr(), self._mod_u[i].data_ptr(), self._mod_v[i].data_ptr()
            s.inv_sigma, s.wv = self.inv_sigma.data_ptr() + 4 * i, self._wv[i].data_ptr()
            s.rows, s.cols = rows, cols
            if need_param_grad:
                b = snb[i]
                b.w, b.u, b.v = w.data_ptr(), self._u[i].data_ptr(), self._v[i].data_ptr()
                b.inv_sigma, b.dw = self.inv_sigma.data_ptr() + 4 * i, self.grads[n + ".weight_orig"].data_ptr()
                b.dot = self._dot.data_ptr() + 4 * i
                b.rows, b.cols = rows, cols
        self._sn_dev = self.upload(sn)
        self._snb_dev = self.upload(snb) if need_param_grad else None
        self.nl = nl

    def param_order(self, names):
        # module registration order per layer: bias, weight_orig (esrgan/discriminator.py:12 +
        # torch spectral_norm.py:171-172 re-registers the weight after the bias)
        out = []
        for n in self.sn_names:
            out += [n + ".bias", n + ".weight_orig"]
        return out

    def forward_buf(self, x_nchw: Optional[torch.Tensor], training: bool) -> Buf:
        """Power iteration (training) -> sigma -> pack W/sigma -> convolutions.  x_nchw None: the
        input is already in A[0].  Returns the logits buffer (one C8 image of width B, slot 0)."""
        L.check(self.lib.srx_sn_power_iter(self._sn_dev.data_ptr(), self.nl, 1 if training else 0,
                                           self.max_rows, self.max_cols, stream_ptr(self.device)),
                "srx_sn_power_iter")
        # keep the u, v this forward used (the hook clones them for the same reason,
        # torch spectral_norm.py:107-110)
        if self.need_param_grad:
            torch._foreach_copy_(self._u, self._mod_u)
            torch._foreach_copy_(self._v, self._mod_v)
        return self.forward_c8(x_nchw)

    def forward_logits(self, x_nchw: Optional[torch.Tensor], training: bool) -> torch.Tensor:
        """forward_buf, returned as a [B,1] fp32 tensor."""
        return self.forward_buf(x_nchw, training).t.view(self.B, 8)[:, 0:1].float()

    def set_output_grad(self, g_logits: torch.Tensor) -> None:
        g = self.gA[-1].t
        g.zero_()
        g.view(self.B, 8)[:, 0] = g_logits.reshape(self.B).to(g.dtype)

    def backward_c8(self) -> None:
        super().backward_c8()
        if self.need_param_grad:
            L.check(self.lib.srx_sn_backward(self._snb_dev.data_ptr(), self.nl, self.unpack_max,
                                             1 if self.deterministic_wgrad else 0,
                                             stream_ptr(self.device)), "srx_sn_backward")

    def input_grad(self) -> torch.Tensor:
        return self.c8_to_nchw((self.gA[0], 0), self.C_in)

    # interface used by host._PlanFn
    training = True

    def forward(self, x_nchw: torch.Tensor) -> torch.Tensor:
        return self.forward_logits(x_nchw, self.training)

    def backward(self, g_logits: torch.Tensor) -> Optional[torch.Tensor]:
        self.set_output_grad(g_logits)
        self.backward_c8()
        return self.input_grad() if self.need_input_grad else None


# ===========================================================================
# VGG19 features[:36]
# ===========================================================================
VGG19_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M",
             512, 512, 512, 512, "M", 512, 512, 512, 512]          # torchvision cfg "E", last pool dropped
VGG19_CONV_IDX = [0, 2, 5, 7, 10, 12, 14, 16, 19, 21, 23, 25, 28, 30, 32, 34]


class VGGFeaturePlan(ChainPlan):
    """relu5_4 features of a frozen VGG19 (utils/loss.py:15-17): forward for an input, and the
    data gradient back to that input (parameters take no gradient)."""

    def __init__(self, params: Dict[str, torch.Tensor], B, H, W, tdtype, *, need_input_grad: bool):
        dev = params["0.weight"].device
        layers, it = [], iter(VGG19_CONV_IDX)
        for c in VGG19_CFG:
            if c == "M":
                layers.append({"kind": "pool"})
            else:
                i = next(it)
                layers.append({"kind": "conv", "w": params[f"{i}.weight"], "b": params[f"{i}.bias"],
                               "wname": f"{i}.weight", "bname": f"{i}.bias", "stride": 1, "act": 0.0})
        if H % 16 or W % 16:
            raise ValueError("VGG feature plan needs H, W divisible by 16 (four 2x2 max-pools)")
        super().__init__(layers, B, 3, H, W, tdtype, dev, need_param_grad=False,
                         need_input_grad=need_input_grad)
