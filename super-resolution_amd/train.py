"""Native training harness for the ESRGAN hot path (counterpart of the reference's
``esrgan/train_esrgan.py:80-118`` step and of its stubbed warm-up loop, :166-171).

Everything between "batch is on the device" and "parameters are updated" runs in
libsrx: generator forward, fused L1 loss + gradient, backward, (data-parallel
gradient exchange), fused Adam on flat parameter storage.  No host
synchronisation happens inside a step; losses accumulate in a device buffer that
the caller reads whenever it wants to log.
"""
from __future__ import annotations

import math
import os
from typing import Callable, Dict, Optional

import torch

from . import _lib as L
from . import chain, engine
from .parallel import broadcast_from_rank0
from .utils import trace
from .utils.flat import flatten_parameters


def cosine_lr(base_lr: float, epoch: int, t_max: int = 1000, eta_min: float = 1e-7) -> float:
    """Closed form of ``CosineAnnealingLR(T_max, eta_min)`` after ``epoch`` scheduler steps
    (esrgan/train_esrgan.py:159-160,188)."""
    return eta_min + (base_lr - eta_min) * (1.0 + math.cos(math.pi * epoch / t_max)) / 2.0


class FlatAdam:
    """torch.optim.Adam(lr, betas=(0.9,0.999)) semantics (esrgan/train_esrgan.py:134-135)
    over flat fp32 parameter / gradient buffers: one srx_adam launch per step."""

    def __init__(self, flat_param: torch.Tensor, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 shapes=None):
        self.p = flat_param
        self.m = torch.zeros_like(flat_param)
        self.v = torch.zeros_like(flat_param)
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        # the learning rate the schedule starts from: torch's schedulers keep it in the parameter
        # group as 'initial_lr', and CosineAnnealingLR(last_epoch=start_epoch-1) — how the reference
        # resumes, esrgan/train_esrgan.py:159-160 — refuses a loaded group without it
        self.initial_lr = lr
        self.step_count = 0
        # per-parameter shapes in parameters() order: what state_dict() needs to speak
        # torch.optim.Adam's format (which indexes parameters by position)
        self.shapes = [tuple(s) for s in shapes] if shapes is not None else [(flat_param.numel(),)]
        assert sum(math.prod(s) for s in self.shapes) == flat_param.numel()

    def state_dict(self) -> dict:
        """The ``torch.optim.Adam.state_dict()`` layout the reference checkpoints hold
        (esrgan/train_esrgan.py:203-204): loadable by a stock Adam over the same parameters."""
        state, off = {}, 0
        if self.step_count > 0:
            for i, shp in enumerate(self.shapes):
                n = math.prod(shp)
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self.m[off:off + n].view(shp).clone(),
                            "exp_avg_sq": self.v[off:off + n].view(shp).clone()}
                off += n
        group = {"lr": self.lr, "initial_lr": self.initial_lr, "betas": tuple(self.betas), "eps": self.eps,
                 "weight_decay": self.wd,
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False,
                 "differentiable": False, "fused": None, "decoupled_weight_decay": False,
                 "params": list(range(len(self.shapes)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd: dict) -> None:
        """Accepts a state_dict written by this class or by ``torch.optim.Adam`` over the same
        parameters (one parameter group, as the reference uses)."""
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self.shapes):
            raise ValueError("optimizer state_dict does not match: expected one group of "
                             f"{len(self.shapes)} parameters")
        g = groups[0]
        self.lr, self.betas, self.eps = g["lr"], tuple(g["betas"]), g["eps"]
        self.wd = g.get("weight_decay", 0.0)
        self.initial_lr = g.get("initial_lr", self.initial_lr)
        self.m.zero_()
        self.v.zero_()
        steps, off = set(), 0
        for i, shp in enumerate(self.shapes):
            n = math.prod(shp)
            st = sd["state"].get(i)
            if st is not None:
                if tuple(st["exp_avg"].shape) != shp:
                    raise ValueError(f"optimizer state for parameter {i} has shape "
                                     f"{tuple(st['exp_avg'].shape)}, expected {shp}")
                self.m[off:off + n].copy_(st["exp_avg"].reshape(-1))
                self.v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
                steps.add(int(float(st["step"])))
            off += n
        if len(steps) > 1:
            raise ValueError("parameters with different step counts cannot share one flat Adam state")
        self.step_count = steps.pop() if steps else 0

    def step(self, flat_grad: torch.Tensor, grad_scale: float = 1.0):
        self.begin_step()
        self.step_range(flat_grad, 0, self.p.numel(), grad_scale)

    def begin_step(self) -> None:
        """Opens an optimizer step that is then applied piecewise with step_range()."""
        self.step_count += 1

    def step_range(self, flat_grad: torch.Tensor, lo: int, hi: int, grad_scale: float = 1.0) -> None:
        """The current step for parameters [lo, hi) only, on the current stream.  Adam is
        elementwise, so a step may be applied range by range, each as soon as its gradients are
        final; every element must be covered exactly once per begin_step()."""
        if hi <= lo:
            return
        L.check(L.lib().srx_adam(self.p.data_ptr() + 4 * lo, flat_grad.data_ptr() + 4 * lo,
                                 self.m.data_ptr() + 4 * lo, self.v.data_ptr() + 4 * lo, hi - lo, self.lr,
                                 self.betas[0], self.betas[1], self.eps, self.wd, self.step_count, grad_scale,
                                 engine.stream_ptr(self.p.device)), "srx_adam")


def _checked_batch(lr_img: torch.Tensor, hr_img: torch.Tensor, lr_shape, hr_shape, device):
    """The kernels read the batch through raw pointers as contiguous fp32 of the shapes the plans
    were built for: anything else (another dtype, size or device) would be read as garbage or out
    of bounds, so it is converted or rejected here."""
    for name, t, shp in (("lr", lr_img, lr_shape), ("hr", hr_img, hr_shape)):
        if tuple(t.shape) != tuple(shp):
            raise ValueError(f"{name} batch has shape {tuple(t.shape)}, the trainer was built for {tuple(shp)}")
        if t.device != device:
            raise ValueError(f"{name} batch is on {t.device}, the trainer runs on {device}")
    return lr_img.float().contiguous(), hr_img.float().contiguous()


def _backward_with_exchange(plan, grad_sync) -> bool:
    """Backward + unpack of a generator plan.  With an exchange object that has the
    begin/ready/finish protocol (parallel.OverlappedGradAllReduce) and a two-stream plan, each
    gradient range is unpacked and handed to the exchange as soon as its weight-gradient
    launches are enqueued; returns True when the exchange has been done that way."""
    if grad_sync is not None and hasattr(grad_sync, "ready") and plan.bwd_seg_ranges is not None:
        grad_sync.begin(plan.flat_grad)
        if plan.run_backward_ops(grad_sync.ready):
            grad_sync.finish()
            return True
    else:
        plan.run_backward_ops()
    plan.run_unpack()
    return False


def _backward_and_step(plan, opt: "FlatAdam") -> bool:
    """Single-process form: backward with the optimizer applied range by range on the side stream,
    each range right after its weight-gradient launches and its unpack — the master parameters
    are not read again before the next forward's packing, which waits for the side stream.
    Returns True if the step has been applied that way (False: the plan has no two-stream form
    and the caller unpacks and steps as usual)."""
    if plan.bwd_segments is None or plan.bwd_seg_ranges is None:
        return False
    opt.begin_step()
    g = plan.flat_grad
    done = plan.run_backward_ops(lambda lo, hi: opt.step_range(g, lo, hi))
    assert done
    return True


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
        if grad_sync is not None:
            broadcast_from_rank0([self.flat], getattr(grad_sync, "group", None))
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
        self.opt = FlatAdam(self.flat, lr=lr, shapes=[p.shape for p in gen.parameters()])
        self.loss_acc = torch.zeros(1, dtype=torch.float32, device=dev)
        self.grad_sync = grad_sync
        self.steps = 0
        self._lr_shape = (B, kw["in_ch"], h, w)
        self._hr_shape = (B, kw["out_ch"], 4 * h, 4 * w)

    # single process: Adam applied range by range beside the rest of backward instead of one launch
    # after it.  Same arithmetic; measured 0.6-1 % slower (the extra short launches sit between the
    # weight-gradient launches of the side stream), so off.
    step_in_backward = False

    def _sub_step(self, p, lr_img, hr_img):
        with trace.range_("generator forward"):
            Y = p.forward_c8(lr_img)
        with trace.range_("backward (+ gradient exchange)"):
            return self._loss_and_backward(p, Y, hr_img)

    def _loss_and_backward(self, p, Y, hr_img):
        L.check(p.lib.srx_l1_c8_nchw(p.dt, Y.ptr(), Y.img_stride, hr_img.data_ptr(), self.loss_acc.data_ptr(),
                                     p.GY.ptr(), p.GY.img_stride, 1.0 / self.n_streams, Y.B, p.out_ch, Y.H, Y.W,
                                     engine.stream_ptr(p.device)), "srx_l1_c8_nchw")
        if self.n_streams == 1:
            if self.grad_sync is None and self.step_in_backward and _backward_and_step(p, self.opt):
                return "stepped"
            return _backward_with_exchange(p, self.grad_sync)
        p.run_backward_ops()
        return False

    def step(self, lr_img: torch.Tensor, hr_img: torch.Tensor) -> None:
        lr_img, hr_img = _checked_batch(lr_img, hr_img, self._lr_shape, self._hr_shape, self.flat.device)
        exchanged = False
        if self.n_streams == 1:
            exchanged = self._sub_step(self.plan, lr_img, hr_img)
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
        if exchanged != "stepped":
            if self.grad_sync is not None and not exchanged:
                with trace.range_("gradient exchange"):
                    self.grad_sync(g)
            with trace.range_("optimizer"):
                self.opt.step(g)
        self.steps += 1

    def mean_loss(self) -> float:
        """Average L1 loss since the last call (one host sync).  Sub-batch losses are each a
        mean over their own pixels, so their sum is divided by the stream count."""
        v = float(self.loss_acc.item()) / max(self.steps, 1) / self.n_streams
        self.loss_acc.zero_()
        self.steps = 0
        return v


class AdversarialTrainer:
    """One discriminator step + one generator step per call, the body of the reference's
    ``train_one_epoch`` (esrgan/train_esrgan.py:87-111), entirely in libsrx:

      fake = G(lr)
      D step : loss_disc = (BCE(D(hr),1) + BCE(D(fake),0)) / 2  -> Adam on D       (:88-98)
      G step : loss_gen = l_l1*L1(fake,hr) + l_adv*BCE(D(fake),1) + l_percep*VGG(fake,hr)
               -> Adam on G                                                          (:102-111)

    As in the reference the discriminator runs three training-mode forwards per step, so
    every spectral-norm layer's u, v advance three power iterations, and the G step sees the
    discriminator weights after their update.  The G step computes no discriminator weight
    gradients (the reference discards them at the next ``disc.zero_grad``, :95).  bf16 needs
    no loss scaling, so there is no GradScaler.  Losses accumulate on the device;
    ``losses()`` is the only host synchronisation.

    grad_sync_d / grad_sync_g (optional) are called with the flat gradient buffer before the
    respective optimizer step (data-parallel all-reduce hooks)."""

    def __init__(self, gen, disc, vgg_loss, batch_shape, compute_dtype=torch.bfloat16, lr: float = 2e-4,
                 lambda_l1: float = 1.0, lambda_adv: float = 5e-3, lambda_percep: float = 1.0,
                 grad_sync_g: Optional[Callable] = None, grad_sync_d: Optional[Callable] = None):
        B, _, h, w = batch_shape
        H, W = 4 * h, 4 * w
        self.gen, self.disc, self.vgg_loss = gen, disc, vgg_loss
        self.lams = (lambda_l1, lambda_adv, lambda_percep)
        self.flat_g = flatten_parameters(gen)
        self.flat_d = flatten_parameters(disc)
        if grad_sync_g is not None or grad_sync_d is not None:
            # replicas start from rank 0's generator, discriminator and spectral-norm u / v (which
            # then evolve identically on every rank and are never communicated again)
            grp = getattr(grad_sync_g if grad_sync_g is not None else grad_sync_d, "group", None)
            broadcast_from_rank0([self.flat_g, self.flat_d] + [b for b in disc.buffers()], grp)
        dev = self.flat_g.device
        self.device = dev
        gparams = {n: p.data for n, p in gen.named_parameters()}
        self.gplan = engine.GeneratorPlan(
            gparams, B, h, w, compute_dtype, in_ch=gen.conv_first.in_channels,
            out_ch=gen.conv_last.out_channels, nf=gen.conv_first.out_channels,
            gc=gen.body[0].rdb1.conv1.out_channels if len(gen.body) else 32, num_blocks=len(gen.body),
            need_grad=True, need_input_grad=False)
        dparams = {n: p.data for n, p in disc.named_parameters()}
        dbufs = {n: b for n, b in disc.named_buffers()}
        mk = lambda pg, ig: chain.DiscriminatorPlan(dparams, dbufs, B, H, W, compute_dtype,
                                                    need_param_grad=pg, need_input_grad=ig)
        # Inside this step a discriminator pass always has another stream's launches beside it (the
        # generator's forward pass, the perceptual branch): its weight-gradient launches take few
        # workgroups, so that those find CUs (a fused dense-block workgroup shares a CU with nothing).
        keep, chain.ChainPlan.wgrad_blocks = chain.ChainPlan.wgrad_blocks, self.disc_wgrad_blocks
        try:
            self.d_real, self.d_fake, self.d_gen = mk(True, False), mk(True, False), mk(False, True)
        finally:
            chain.ChainPlan.wgrad_blocks = keep
        vparams = {n: (p.data if p.dtype == torch.float32 else p.data.float())
                   for n, p in vgg_loss.vgg.named_parameters()}
        self._vparams = vparams
        self.v_fake = chain.VGGFeaturePlan(vparams, B, H, W, compute_dtype, need_input_grad=True)
        self.v_real = chain.VGGFeaturePlan(vparams, B, H, W, compute_dtype, need_input_grad=False)
        self.v_fake.run_packing()
        self.v_real.run_packing()                 # frozen weights: packed once
        self.opt_g = FlatAdam(self.flat_g, lr=lr, shapes=[p.shape for p in gen.parameters()])
        self.opt_d = FlatAdam(self.flat_d, lr=lr, shapes=[p.shape for p in disc.parameters()])
        self.acc = torch.zeros(4, dtype=torch.float32, device=dev)   # loss_disc, adv, l1, vgg
        self.grad_sync_g, self.grad_sync_d = grad_sync_g, grad_sync_d
        self.steps = 0
        self.B = B
        self._ev_in, self._ev_real, self._ev_y, self._ev_vgg = (torch.cuda.Event() for _ in range(4))
        engine.side_stream(dev)      # created here, not inside the first step

    step_in_backward = False    # as in GeneratorWarmupTrainer

    # workgroups per weight-gradient launch of the discriminator plans (chain.ChainPlan.wgrad_blocks;
    # 512 for a discriminator that runs alone).  A/B on one box, per step: HR 128 x 128 (config 3)
    # 7.22 ms at 512, 6.60 at 256, 6.30 at 128, 6.25 at 96, 6.7 at 32; HR 256 x 256 16.4 at 512, 15.7 at 256,
    # 15.5 at 192 and 128, 16.6 at 64.
    disc_wgrad_blocks = int(os.environ.get("SRX_DISC_WGRAD_BLOCKS", "128"))

    # D(hr) pass and VGG(hr) features beside the generator forward, the perceptual branch beside
    # the discriminator step and the adversarial branch
    overlap_real_pass = True

    def set_lr(self, lr: float) -> None:
        self.opt_g.lr = lr
        self.opt_d.lr = lr

    def _acc_ptr(self, i: int) -> int:
        return self.acc.data_ptr() + 4 * i

    def _d_pass(self, plan, x_nchw, fake_c8, target: float) -> None:
        lib, sp = plan.lib, engine.stream_ptr(self.device)
        if fake_c8 is not None:
            plan.A[0].t.copy_(fake_c8.t)           # fake.detach(): the same values, no graph
        out = plan.forward_buf(x_nchw, training=True)
        L.check(lib.srx_bce_logits(plan.dt, out.ptr(), self.B, target, 0.5, self._acc_ptr(0),
                                   plan.gA[-1].ptr(), 1.0, sp), "srx_bce_logits")
        plan.backward_c8()

    def _vgg_branch(self, Y, fb, l_percep: float) -> None:
        """VGGLoss(fake, hr) (utils/loss.py:21) and its gradient w.r.t. fake, on the current stream."""
        vf = self.v_fake
        vf.A[0].t.copy_(Y.t)
        fa = vf.forward_c8(None, pack=False)
        Bf, Cf, Hf, Wf = vf.geo[-1]
        g = vf.gA[-1]
        L.check(vf.lib.srx_l1_c8(vf.dt, fa.ptr(), fa.img_stride, fb.ptr(), fb.img_stride, self._acc_ptr(3),
                                 g.ptr(), g.img_stride, l_percep, 0.0, Bf, Cf, Hf, Wf,
                                 engine.stream_ptr(self.device)), "srx_l1_c8")
        vf.backward_c8()

    def step(self, lr_img: torch.Tensor, hr_img: torch.Tensor) -> None:
        gp, lib = self.gplan, self.gplan.lib
        lr_img, hr_img = _checked_batch(lr_img, hr_img, (gp.B, gp.in_ch, gp.h, gp.w),
                                        (gp.B, gp.out_ch, 4 * gp.h, 4 * gp.w), self.device)
        sp = engine.stream_ptr(self.device)
        l_l1, l_adv, l_percep = self.lams
        vf, vr = self.v_fake, self.v_real
        trace.push("generator forward + D(hr), VGG(hr)")
        if self.overlap_real_pass:
            # Everything that needs only the HR batch — D(hr) forward + backward and the VGG
            # features of hr — runs on the side stream under the generator's forward pass (a chain
            # of short launches that leaves most of the chip idle).  The order of the three
            # spectral-norm updates (real, fake, generator step) is kept by the event below.
            main, side = torch.cuda.current_stream(self.device), engine.side_stream(self.device)
            self._ev_in.record(main)
            # the generator's launches are enqueued first: the host needs ~3.5 us per launch, and
            # the longer chain must not wait for the shorter one to be enqueued
            Y = gp.forward_c8(lr_img)
            side.wait_event(self._ev_in)
            with torch.cuda.stream(side):
                self._d_pass(self.d_real, hr_img, None, 1.0)
                fb = vr.forward_c8(hr_img, pack=False)
                self._ev_real.record(side)
            main.wait_event(self._ev_real)
        else:
            Y = gp.forward_c8(lr_img)
            # ---- discriminator step (esrgan/train_esrgan.py:89-98)
            self._d_pass(self.d_real, hr_img, None, 1.0)
            fb = vr.forward_c8(hr_img, pack=False)
        trace.pop()
        trace.push("discriminator step")
        if self.overlap_real_pass:
            # the perceptual branch (VGG(fake) forward, L1 against VGG(hr), data gradient back to
            # fake) needs only Y: beside the discriminator step and the adversarial branch
            self._ev_y.record(main)
            side.wait_event(self._ev_y)
            with torch.cuda.stream(side):
                self._vgg_branch(Y, fb, l_percep)
                self._ev_vgg.record(side)
        self._d_pass(self.d_fake, None, Y, 0.0)
        gd = self.d_real.flat_grad
        L.check(lib.srx_axpby(L.F32, self.d_fake.flat_grad.data_ptr(), gd.data_ptr(), 1.0, 1.0, gd.numel(), sp),
                "srx_axpby")
        if self.grad_sync_d is not None:
            self.grad_sync_d(gd)
        self.opt_d.step(gd)
        trace.pop()
        trace.push("generator step")
        # ---- generator step (:102-111)
        dg = self.d_gen
        dg.A[0].t.copy_(Y.t)
        out = dg.forward_buf(None, training=True)
        L.check(lib.srx_bce_logits(dg.dt, out.ptr(), self.B, 1.0, 1.0, self._acc_ptr(1), dg.gA[-1].ptr(),
                                   l_adv, sp), "srx_bce_logits")
        dg.backward_c8()
        if self.overlap_real_pass:
            main.wait_event(self._ev_vgg)
        else:
            self._vgg_branch(Y, fb, l_percep)
        L.check(lib.srx_l1_c8_nchw(gp.dt, Y.ptr(), Y.img_stride, hr_img.data_ptr(), self._acc_ptr(2),
                                   gp.GY.ptr(), gp.GY.img_stride, l_l1, Y.B, gp.out_ch, Y.H, Y.W, sp),
                "srx_l1_c8_nchw")
        n = gp.GY.t.numel()
        for src in (dg.gA[0], vf.gA[0]):           # dL/dfake = sum of the three branches
            L.check(lib.srx_axpby(gp.dt, src.ptr(), gp.GY.ptr(), 1.0, 1.0, n, sp), "srx_axpby")
        if self.grad_sync_g is None and self.step_in_backward and _backward_and_step(gp, self.opt_g):
            pass                                     # Adam went out range by range beside backward
        else:
            exchanged = _backward_with_exchange(gp, self.grad_sync_g)
            gg = gp.flat_grad
            if self.grad_sync_g is not None and not exchanged:
                self.grad_sync_g(gg)
            self.opt_g.step(gg)
        trace.pop()
        self.steps += 1

    def losses(self) -> Dict[str, float]:
        """Mean losses since the last call, named as the reference logs them
        (esrgan/train_esrgan.py:113-117).  One host synchronisation."""
        a = (self.acc / max(self.steps, 1)).tolist()
        self.acc.zero_()
        self.steps = 0
        l_l1, l_adv, l_percep = self.lams
        return {"loss_disc": a[0], "gen_adv_loss": a[1], "gen_l1_loss": a[2], "gen_vgg_loss": a[3],
                "loss_gen": l_l1 * a[2] + l_adv * a[1] + l_percep * a[3]}
