"""Plan management shared by the drop-in nn.Module classes: choice of compute dtype,
the autograd node that runs a recorded plan, gradient publication into ``param.grad``."""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
from torch import nn


def _auto_compute_dtype(module: nn.Module, x: torch.Tensor) -> torch.dtype:
    want = getattr(module, "compute_dtype", None)
    if want is not None:
        return want
    p = next(module.parameters())
    if torch.is_autocast_enabled() or p.dtype in (torch.float16, torch.bfloat16) or \
            x.dtype in (torch.float16, torch.bfloat16):
        return torch.bfloat16
    return torch.float32


class ParamTracker:
    """name -> fp32 tensor whose storage a module's plans read, kept current.

    Plans hold raw pointers into these tensors.  ``module.to()`` / ``.half()`` and
    ``flatten_parameters`` (building a trainer) give the parameters new storage; the pointer
    signature taken at every call notices that, and ``changed`` then tells the owner to drop its
    plans.  fp32 parameters are used in place (in-place optimizer updates are seen by the next
    weight packing); low-precision parameters are mirrored into fp32 copies refreshed per call."""

    def __init__(self):
        self.sig = None
        self.p32: Optional[Dict[str, torch.Tensor]] = None

    def current(self, module: nn.Module):
        """Returns (params, changed)."""
        named = list(module.named_parameters())
        sig = tuple(p.data_ptr() for _, p in named) + tuple(b.data_ptr() for b in module.buffers())
        changed = sig != self.sig
        if changed:
            self.sig, self.p32 = sig, None
        if all(p.dtype == torch.float32 for _, p in named):
            self.p32 = {n: p.data for n, p in named}
        elif self.p32 is None:
            self.p32 = {n: p.data.float().contiguous() for n, p in named}
        else:
            for n, p in named:
                self.p32[n].copy_(p.data)
        return self.p32, changed


class _PlanFn(torch.autograd.Function):
    """Autograd node for one plan execution (whole module forward/backward)."""

    @staticmethod
    def forward(ctx, x, anchor, host):
        plan = host._acquire_plan(x, need_grad=True, need_input_grad=x.requires_grad)
        out = plan.forward(x)
        ctx.plan, ctx.host, ctx.version = plan, host, plan.version
        ctx.in_dtype = x.dtype
        return out if x.dtype == torch.float32 else out.to(x.dtype)

    @staticmethod
    def backward(ctx, gout):
        plan, host = ctx.plan, ctx.host
        if plan.version != ctx.version:
            raise RuntimeError("libsrx: the activations saved for this backward were overwritten by a "
                               "later forward of the same module (more than 2 forwards in flight)")
        acc_flat = host._grads_alias_plan(plan)
        old = plan.flat_grad.clone() if acc_flat else None
        gx = plan.backward(gout)
        if old is not None:
            plan.flat_grad.add_(old)
        host._publish_grads(plan)
        plan.busy = False
        if gx is not None and ctx.in_dtype != torch.float32:
            gx = gx.to(ctx.in_dtype)
        return gx, None, None


class _SrxHost(nn.Module):
    """Plan management shared by the four module classes."""

    compute_dtype: Optional[torch.dtype] = None     # None: fp32 unless autocast / low-precision params
    _MAX_PLANS = 2

    def _plan_kwargs(self) -> dict:          # pragma: no cover - overridden
        raise NotImplementedError

    def _build_plan(self, params, B, h, w, tdtype, need_grad, need_input_grad):   # pragma: no cover
        raise NotImplementedError

    def _srx_state(self) -> dict:
        st = self.__dict__.get("_srx")
        if st is None:
            st = {"plans": {}, "tracker": ParamTracker()}
            self.__dict__["_srx"] = st
        return st

    def _apply(self, fn, *a, **k):
        self.__dict__.pop("_srx", None)       # parameters moved / cast: drop plans (they hold pointers)
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        return super().load_state_dict(*a, **k)   # copies in place: pointers stay valid

    def _params32(self) -> Dict[str, torch.Tensor]:
        """name -> fp32 tensor whose storage the plans read.  fp32 parameters are used in
        place; low-precision parameters are mirrored into fp32 copies refreshed per forward."""
        st = self._srx_state()
        params, changed = st["tracker"].current(self)
        if changed:
            st["plans"].clear()
        return params

    def _acquire_plan(self, x: torch.Tensor, need_grad: bool, need_input_grad: bool):
        if x.device.type != "cuda":
            raise RuntimeError("libsrx modules run on a ROCm GPU only (input is on %s): the HIP extension "
                               "is the only implementation" % x.device)
        if x.dim() != 4:
            raise ValueError("expected a 4-D NCHW input")
        params = self._params32()
        st = self._srx_state()
        tdtype = _auto_compute_dtype(self, x)
        key = (tuple(x.shape), tdtype, need_grad, need_input_grad)
        pool: List = st["plans"].setdefault(key, [])
        plan = None
        for p in pool:
            if not p.busy:
                plan = p
                break
        if plan is None:
            if len(pool) < self._MAX_PLANS:
                B, _, h, w = x.shape
                plan = self._build_plan(params, B, h, w, tdtype, need_grad, need_input_grad)
                plan.version, plan.busy = 0, False
                pool.append(plan)
            else:
                plan = min(pool, key=lambda p: p.version)
        plan.version += 1
        plan.busy = need_grad
        plan.training = self.training
        return plan

    # ------------------------------------------------------------ gradients
    def _grads_alias_plan(self, plan) -> bool:
        name0 = plan.param_names[0]
        p0 = dict(self.named_parameters())[name0]
        return p0.grad is not None and p0.grad.data_ptr() == plan.grads[name0].data_ptr()

    def _publish_grads(self, plan) -> None:
        for n, p in self.named_parameters():
            if not p.requires_grad:
                continue
            g = plan.grads[n]
            if p.grad is None:
                p.grad = g if p.dtype == torch.float32 else g.to(p.dtype)
            elif p.grad.data_ptr() != g.data_ptr():
                p.grad.add_(g.to(p.grad.dtype))

    def _out_shape(self, shape):          # pragma: no cover - overridden
        """Output shape for an NCHW input shape; raises on a channel count the module was not built for."""
        raise NotImplementedError

    @staticmethod
    def _check_channels(shape, expected: int) -> None:
        if len(shape) != 4:
            raise ValueError("expected a 4-D NCHW input")
        if shape[1] != expected:
            raise RuntimeError(f"expected input with {expected} channels, got {shape[1]} (shape {tuple(shape)})")

    def _run(self, x: torch.Tensor) -> torch.Tensor:
        need_grad = torch.is_grad_enabled() and (
            x.requires_grad or any(p.requires_grad for p in self.parameters()))
        out_shape = self._out_shape(tuple(x.shape))
        if x.shape[0] == 0:
            # an empty batch launches nothing; like the stock layers it yields an empty result through
            # which zero gradients reach the input and every parameter
            if x.device.type != "cuda":
                raise RuntimeError("libsrx modules run on a ROCm GPU only (input is on %s)" % x.device)
            out = torch.zeros(out_shape, dtype=x.dtype, device=x.device)
            if need_grad:
                out = out + 0 * (x.sum() + sum(p.sum() for p in self.parameters() if p.requires_grad)).to(x.dtype)
            return out
        if need_grad:
            anchor = torch.empty(0, device=x.device, requires_grad=True)
            return _PlanFn.apply(x, anchor, self)
        plan = self._acquire_plan(x, need_grad=False, need_input_grad=False)
        out = plan.forward(x)
        return out if x.dtype == torch.float32 else out.to(x.dtype)


