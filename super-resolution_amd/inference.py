"""Inference engine for the ESRGAN generator, counterpart of the reference's
``SuperResolutionInference`` (``inference.py:104-230``) for its ``esrgan`` registry entry.

Whole-image inference runs the generator's recorded forward plan once.  Tiled inference
keeps the reference's tile grid (stride = tile - overlap, reflect padding on the bottom /
right so the last full tile ends on the padded edge, clipped tiles past it) and its
window-weighted blend, but runs on the device end to end: one gather kernel cuts all
equally sized tiles of an image straight into the generator's input layout, the tiles go
through the generator in batches instead of one call per tile, and one blend kernel
sums, normalises, crops and clamps.  The reference's blend slices the wrong dimensions of
its window for clipped tiles (``inference.py:225-226``); here the window is sliced on its
spatial dimensions, which is what the loop intends.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib as L
from . import engine
from .host import ParamTracker

SCALE = 4


def load_generator_state_dict(path) -> Dict[str, torch.Tensor]:
    """Read any of the checkpoint layouts the reference writes and unwrap the generator's
    state_dict (inference.py:133-138): a raw state_dict, ``{'model_state_dict': ...}`` or the
    trainer's ``{'gen_state_dict': ...}``.  Loaded with ``weights_only=True``."""
    sd = torch.load(os.fspath(path), map_location="cpu", weights_only=True)
    if isinstance(sd, dict) and "model_state_dict" in sd:
        sd = sd["model_state_dict"]
    elif isinstance(sd, dict) and "gen_state_dict" in sd:
        sd = sd["gen_state_dict"]
    return sd


def blend_window(n: int, kind: str = "hann") -> torch.Tensor:
    """"hann": ``torch.hann_window(n, periodic=False)`` as in inference.py:208-209 (zero at
    both ends, so the outermost output rows / columns get weight 0).  "hann_pos": the interior
    n samples of an (n+2)-point Hann window: strictly positive."""
    if kind == "hann":
        return torch.hann_window(n, periodic=False, dtype=torch.float32)
    if kind == "hann_pos":
        return torch.hann_window(n + 2, periodic=False, dtype=torch.float32)[1:-1].clone()
    raise ValueError(f"unknown window {kind!r}")


class SuperResolutionInference:
    """``SuperResolutionInference('esrgan', checkpoint_path, half_precision, device)``.

    model: an already constructed ``GeneratorESRGAN`` (this package's), or None to build one
    with ``num_features`` / ``num_blocks`` and load ``checkpoint_path`` into it with
    ``strict=True`` (inference.py:140-141).  half_precision selects bf16 MFMA compute (the
    reference's ``.half()`` + autocast); otherwise fp32.
    """

    def __init__(self, model_name: str = "esrgan", checkpoint_path=None, half_precision: bool = False,
                 device: Optional[str] = None, *, model=None, num_features: int = 64, num_blocks: int = 23,
                 tile_batch: int = 12, window: str = "hann"):
        if model_name != "esrgan":
            raise ValueError(f"Unknown model: '{model_name}'. Choose from ['esrgan']")
        device = device or "cuda"
        if not str(device).startswith("cuda"):
            raise RuntimeError("the HIP extension is the only implementation: device must be a ROCm GPU")
        self.device = torch.device(device)
        self.half = bool(half_precision)
        if model is None:
            from .esrgan.generator import GeneratorESRGAN
            model = GeneratorESRGAN(num_features=num_features, num_blocks=num_blocks)
            if checkpoint_path is None:
                raise FileNotFoundError("no checkpoint given: pass checkpoint_path or a constructed model")
            model.load_state_dict(load_generator_state_dict(checkpoint_path), strict=True)
        self.model = model.to(self.device)
        self.model.eval()
        self.tdtype = torch.bfloat16 if self.half else torch.float32
        self.tile_batch = max(1, int(tile_batch))
        self.window_kind = window
        self._plans: Dict[Tuple[int, int, int], engine.GeneratorPlan] = {}
        self._tracker = ParamTracker()
        self._windows: Dict[int, torch.Tensor] = {}
        self.lib = L.lib()

    # ------------------------------------------------------------------ plans
    def _plan(self, B: int, h: int, w: int) -> engine.GeneratorPlan:
        g = self.model
        # fp32 master copies the plans read (weights are re-packed from them every forward, so
        # in-place updates of fp32 parameters are picked up); parameters that moved to new storage
        # (model.to() / .half(), a trainer's flatten_parameters) invalidate the plans
        self._params, moved = self._tracker.current(g)
        if moved:
            self._plans.clear()
        key = (B, h, w)
        plan = self._plans.get(key)
        if plan is None:
            plan = engine.GeneratorPlan(
                self._params, B, h, w, self.tdtype, in_ch=g.conv_first.in_channels,
                out_ch=g.conv_last.out_channels, nf=g.conv_first.out_channels,
                gc=g.body[0].rdb1.conv1.out_channels if len(g.body) else 32, num_blocks=len(g.body),
                need_grad=False, need_input_grad=False)
            if len(self._plans) >= 8:             # plans own activation buffers: keep a few shapes
                self._plans.pop(next(iter(self._plans)))
            self._plans[key] = plan
        return plan

    def _sp(self) -> int:
        return engine.stream_ptr(self.device)

    # ------------------------------------------------------------- inference
    @torch.no_grad()
    def upscale(self, lr_tensor: torch.Tensor, tile_size: int = 0, tile_overlap: int = 32,
                clamp: bool = True) -> torch.Tensor:
        """[b,c,h,w] float tensor in [0,1] -> [b,c,4h,4w] fp32 on the device (clamped to [0,1]
        as ``run`` does, inference.py:163)."""
        lr_tensor = lr_tensor.to(self.device, torch.float32).contiguous()
        b, c, h, w = lr_tensor.shape
        use_tiling = tile_size > 0 and (h > tile_size or w > tile_size)      # inference.py:151
        if not use_tiling:
            out = self._plan(b, h, w).forward(lr_tensor)                      # inference.py:166-168
            return out.clamp_(0, 1) if clamp else out
        if tile_overlap >= tile_size:
            raise ValueError("tile_overlap must be smaller than tile_size")
        return self._infer_tiled_batch(lr_tensor, tile_size, tile_overlap, clamp)

    def _infer_tiled(self, img: torch.Tensor, tile: int, overlap: int, clamp: bool,
                     want_u8: bool = False) -> torch.Tensor:
        """One [c,h,w] image through the tile grid of inference.py:189-230."""
        return self._infer_tiled_batch(img.unsqueeze(0), tile, overlap, clamp, want_u8)[0]

    def _infer_tiled_batch(self, imgs: torch.Tensor, tile: int, overlap: int, clamp: bool,
                           want_u8: bool = False) -> torch.Tensor:
        """[b,c,h,w] images through the tile grid of inference.py:189-230.  The grid is the
        reference's (one tile every ``stride`` pixels of the reflect-padded image, clipped past
        the padded edge).  A tile that starts at or beyond the image's own last row / column
        covers padding only, which the final crop (inference.py:230) discards: it is not
        computed.  The remaining tiles of ALL images go through the generator in equally sized
        batches per tile shape, and the last convolution of each batch writes straight into the
        tile store the blend kernel reads."""
        nimg, c, h, w = imgs.shape
        stride = tile - overlap
        pad_h = max(0, (stride - (h - tile) % stride) % stride)               # inference.py:195-196
        pad_w = max(0, (stride - (w - tile) % stride) % stride)
        if pad_h >= h or pad_w >= w:
            raise ValueError("reflect padding needs pad < image size (image much smaller than the tile)")
        ph, pw = h + pad_h, w + pad_w
        ys, xs = list(range(0, ph, stride)), list(range(0, pw, stride))
        ny, nx = len(ys), len(xs)
        ntile = ny * nx
        # per-image tile tables (the blend kernel's [ny][nx] grid), all in one host array
        tables = (L.Tile * (nimg * ntile))()
        by_shape: Dict[Tuple[int, int], List[Tuple[int, int]]] = {}
        for i in range(nimg):
            for ky, y in enumerate(ys):
                for kx, x in enumerate(xs):
                    t = tables[i * ntile + ky * nx + kx]
                    t.y0, t.x0 = y, x
                    if y < h and x < w:
                        t.th, t.tw = min(y + tile, ph) - y, min(x + tile, pw) - x
                        by_shape.setdefault((t.th, t.tw), []).append((i, ky * nx + kx))
                    else:
                        t.th = t.tw = 0          # covers cropped padding only: never read by the blend
        # batches: per shape, as few as the batch limit allows, equally sized; store offsets in
        # batch order so that a batch's outputs are one contiguous block
        batches: List[Tuple[int, int, List[Tuple[int, int]]]] = []
        off = 0
        for (th, tw), jobs in by_shape.items():
            nb = -(-len(jobs) // self.tile_batch)
            per = -(-len(jobs) // nb)
            n_el = th * SCALE * tw * SCALE * 8
            for s_ in range(0, len(jobs), per):
                part = jobs[s_:s_ + per]
                for (i, k) in part:
                    tables[i * ntile + k].offset = off
                    off += n_el
                batches.append((th, tw, part))
        store = torch.empty(off, dtype=self.tdtype, device=self.device)
        # gather tables: the jobs of each batch in order (one gather launch per run of one image)
        order = [(i, k) for _, _, part in batches for (i, k) in part]
        gtab = (L.Tile * len(order))(*[tables[i * ntile + k] for (i, k) in order])
        raw = C.string_at(C.addressof(tables), C.sizeof(tables)) + C.string_at(C.addressof(gtab), C.sizeof(gtab))
        tab_dev = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)
        gtab_ptr = tab_dev.data_ptr() + C.sizeof(tables)
        tsz = C.sizeof(L.Tile)
        dt = engine.dtype_id(self.tdtype)
        esize = store.element_size()
        pos = 0
        for th, tw, part in batches:
            plan = self._plan(len(part), th, tw)
            plan.run_packing()
            j = 0
            while j < len(part):                     # runs of one image: one gather launch each
                j1 = j
                while j1 + 1 < len(part) and part[j1 + 1][0] == part[j][0]:
                    j1 += 1
                L.check(self.lib.srx_tile_gather(dt, imgs[part[j][0]].data_ptr(), c, h, w, gtab_ptr + (pos + j) * tsz,
                                                 j1 - j + 1, th, tw,
                                                 plan.X0.ptr() + j * plan.X0.img_stride * esize,
                                                 plan.X0.img_stride, self._sp()), "srx_tile_gather")
                j = j1 + 1
            # the plan's last convolution (conv_last) writes this batch's block of the store
            last = plan.fwd_ops[len(plan.fwd_ops) - 1].u.conv
            own_y = last.y
            last.y = store.data_ptr() + tables[part[0][0] * ntile + part[0][1]].offset * esize
            try:
                plan.run(plan.fwd_ops)
            finally:
                last.y = own_y
            pos += len(part)
        win = self._windows.get(tile)
        if win is None:
            win = blend_window(tile * SCALE, self.window_kind).to(self.device)
            self._windows[tile] = win
        OH, OW = h * SCALE, w * SCALE
        out = torch.empty((nimg, c, OH, OW), dtype=torch.uint8 if want_u8 else torch.float32, device=self.device)
        for i in range(nimg):
            L.check(self.lib.srx_tile_blend(dt, store.data_ptr(), tab_dev.data_ptr() + i * ntile * tsz, ny, nx, stride,
                                            tile, SCALE, win.data_ptr(),
                                            None if want_u8 else out[i].data_ptr(),
                                            out[i].data_ptr() if want_u8 else None, c, OH, OW,
                                            1 if clamp else 0, self._sp()), "srx_tile_blend")
        self._keep = (tab_dev, store)               # stay alive until the stream is done with them
        return out

    @torch.no_grad()
    def run(self, lr_image, tile_size: int = 0, tile_overlap: int = 32):
        """PIL image in, PIL image out (inference.py:144-164).  ``to_tensor`` scaling to [0,1]
        and ``to_pil_image``'s ``mul(255).byte()`` are reproduced without torchvision."""
        import numpy as np
        from PIL import Image
        arr = np.asarray(lr_image.convert("RGB"), dtype=np.uint8)
        lr = torch.from_numpy(arr).permute(2, 0, 1).float().div(255.0).unsqueeze(0)
        sr = self.upscale(lr, tile_size, tile_overlap)[0]
        u8 = sr.mul(255).byte().permute(1, 2, 0).cpu().numpy()
        return Image.fromarray(u8, mode="RGB")
