"""Host-side execution plans for the ESRGAN hot path.

A *plan* is built once per (module, batch shape, dtype, grad-mode).  It owns the
C8 activation buffers, the packed-weight buffers and three recorded op lists
(forward, backward, weight-packing) whose descriptors hold raw device pointers;
running a plan is one ``srx_run_ops`` call per phase.  PyTorch supplies device
memory and the stream; all arithmetic runs in libsrx (HIP).

Dense blocks (esrgan/generator.py:6-23) are executed without concatenation:
each block owns one (nf+4*gc)-channel buffer; convolution k reads channel
planes [0, nf+(k-1)gc) and writes planes [nf+(k-1)gc, nf+k*gc); conv5 writes
``0.2*y + x`` straight into the next block's first nf planes.

The data gradient of a dense block is run as the mirror image: a "gradient
buffer" [gy1|gy2|gy3|gy4|g_out] per block, where the gradient of growth slice
j is ONE convolution over the planes [gy_{j+1} .. g_out] with the stacked,
transposed, tap-flipped weights of convolutions j+1..5 (so no read-modify-write
accumulation is needed), masked by LeakyReLU'(x_j) from the saved forward
buffer.  Weight gradients of the five convolutions are one launch per block.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib as L

LRELU = 0.2
RES = 0.2


def _ceil(a: int, b: int) -> int:
    return (a + b - 1) // b


def dtype_id(t: torch.dtype) -> int:
    if t == torch.bfloat16:
        return L.BF16
    if t == torch.float32:
        return L.F32
    raise ValueError(f"compute dtype must be torch.bfloat16 or torch.float32, got {t}")


def stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


_SIDE_STREAMS: Dict[int, "torch.cuda.Stream"] = {}


def side_stream(device) -> "torch.cuda.Stream":
    """The device's stream for work that runs BESIDE the current stream's launch chain (weight
    gradients next to data gradients, packing under the forward pass).  NORMAL priority, like the
    caller's stream: on this runtime a stream that waits for an event of a LOWER-priority stream can
    keep that stream from ever being scheduled (observed: a high-priority stream waiting for the
    main stream, and the main stream waiting for a low-priority side stream while other processes
    share the card; tools/stream_priority_probe.py), and a step's streams wait for each other in
    both directions.  With equal priorities the hardware scheduler time-slices and every wait ends.
    The runtime gives each of the first four streams of a priority its own hardware queue, so this
    stream is created when the first plan or trainer is built (before a process group's streams):
    a side stream that shares the main chain's queue runs after it instead of beside it (13.2
    instead of 11.1 ms per step).  Created through the library (srx_stream_create)."""
    idx = torch.device(device).index
    idx = torch.cuda.current_device() if idx is None else idx
    st = _SIDE_STREAMS.get(idx)
    if st is None:
        prio = int(os.environ.get("SRX_SIDE_STREAM_PRIORITY", "0"))     # A/B runs only
        out = C.c_void_p()
        with torch.cuda.device(idx):
            L.check(L.lib().srx_stream_create(prio, C.byref(out)), "srx_stream_create")
        st = torch.cuda.ExternalStream(out.value, device=torch.device("cuda", idx))
        _SIDE_STREAMS[idx] = st
    return st


class Buf:
    """A C8 activation buffer [B][C8][H][W][8]."""

    def __init__(self, B: int, C_: int, H: int, W: int, tdtype, device, zero: bool = False):
        self.B, self.C, self.H, self.W = B, C_, H, W
        self.C8 = _ceil(C_, 8)
        self.plane = H * W * 8
        self.img_stride = self.C8 * self.plane
        n = B * self.img_stride
        self.t = (torch.zeros if zero else torch.empty)(n, dtype=tdtype, device=device)
        self.esize = self.t.element_size()

    def ptr(self, oct0: int = 0) -> int:
        return self.t.data_ptr() + oct0 * self.plane * self.esize

    @property
    def nbytes(self) -> int:
        return self.t.numel() * self.esize


View = Tuple[Buf, int]   # (buffer, first octet plane)


class OpList(list):
    """List of L.Op plus, per op, (kernel name, algorithmic FLOPs) for the benchmark."""

    def __init__(self):
        super().__init__()
        self.meta: List[Tuple[str, float]] = []


def conv_kernel_name(dt: int, ksize: int, stride: int, OH: int, OW: int, cout: int, B: int = 1 << 20) -> str:
    """Mirror of the instantiation choice in csrc/conv.hip (dispatch<T>)."""
    t = "bf16" if dt == L.BF16 else "float"
    nb = 2 if (_ceil(cout, 32) % 2 == 0) else 1

    def nb_small(ltw: int) -> int:
        tw, th = 1 << ltw, 4 * (32 >> ltw)
        if nb == 2 and os.environ.get("SRX_CONV_SMALL_NB1", "1") != "0" and \
                _ceil(OW, tw) * _ceil(OH, th) * B * (_ceil(cout, 32) // 2) < 256:
            return 1
        return nb
    if ksize == 1:
        cfg = (1, nb, 1, 5, 1)
    elif stride == 1:
        if OW >= 24:
            wgs8 = _ceil(OW, 32) * _ceil(OH, 8) * B * (_ceil(cout, 32) // nb)
            cfg = (1, 1, 4, 5, 3) if (OH * OW >= 128 * 128 and nb == 1) else (1, nb, 1 if wgs8 < 256 else 2, 5, 3)
        elif OW >= 12:
            cfg = (1, nb_small(4), 1, 4, 3)
        else:
            cfg = (1, nb_small(3), 1, 3, 3)
    elif OW >= 24:
        cfg = (2, nb, 1, 5, 3)
    else:
        ltw = 4 if OW >= 12 else 3
        cfg = (2, nb_small(ltw), 1, ltw, 3)
    return "conv_kernel<%s,%d,%d,%d,%d,%d>" % ((t,) + cfg)


class Plan:
    """Recorder of ops + owner of buffers.  Subclasses build the op lists."""

    def __init__(self, device, tdtype):
        if device.type != "cuda":
            raise RuntimeError("libsrx plans run on a ROCm device only (got %s); there is no CPU path" % device)
        self.device = device
        self.tdtype = tdtype
        self.dt = dtype_id(tdtype)
        self.esize = 2 if self.dt == L.BF16 else 4
        self.lib = L.lib()
        self.deterministic_wgrad = bool(type(self).deterministic_wgrad)    # fixed when the plan is built
        self._item_by_src = {}
        self.keep: List[torch.Tensor] = []
        self.pack_layers: List[L.PackLayer] = []
        self.packed_tensors: List[Tuple[torch.Tensor, Optional[torch.Tensor]]] = []   # (weights, bias) per layer
        self.pack_segs: List[L.PackSeg] = []
        self.pack_scale_ptrs: List[int] = []
        self.pack_max_elems = 0
        self._pack_dev = None
        self.unpack_items: List[L.UnpackItem] = []
        self.unpack_max = 0
        self._unpack_dev = None
        self.bytes_alloc = 0

    # ------------------------------------------------------------ memory
    def buf(self, B, C_, H, W, zero=False) -> Buf:
        b = Buf(B, C_, H, W, self.tdtype, self.device, zero)
        self.bytes_alloc += b.nbytes
        self.keep.append(b.t)      # op descriptors hold raw pointers: the plan owns the memory
        return b

    def f32(self, n: int, zero=True) -> torch.Tensor:
        t = (torch.zeros if zero else torch.empty)(max(n, 1), dtype=torch.float32, device=self.device)
        self.keep.append(t)
        self.bytes_alloc += t.numel() * 4
        return t

    def upload(self, arr) -> torch.Tensor:
        """Copy a ctypes array to the device; returns the owning uint8 tensor."""
        raw = C.string_at(C.addressof(arr), C.sizeof(arr))
        t = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)
        self.keep.append(t)
        return t

    # ----------------------------------------------------------- packing
    def packed_layer(self, cout: int, cin: int, taps: int, perm: int,
                     segs: Sequence[dict], bias_ptr: int = 0, with_bias: bool = True,
                     dst_w: Optional[torch.Tensor] = None, dst_bias: Optional[torch.Tensor] = None):
        """Register a virtual conv layer to pack; returns (w_ptr, bias_ptr).  dst_w / dst_bias:
        where to pack it (slices of a caller-owned tensor: the fused dense block wants its five
        layers back to back); by default the layer gets tensors of its own."""
        nblk, nchunk = _ceil(cout, 32), _ceil(cin, 16)
        nel = nblk * nchunk * taps * 512
        if dst_w is not None:
            assert dst_w.numel() == nel and dst_w.dtype == self.tdtype and dst_w.is_contiguous()
            w = dst_w
        else:
            w = torch.empty(nel, dtype=self.tdtype, device=self.device)
            self.keep.append(w)
            self.bytes_alloc += nel * self.esize
        bias = None
        if with_bias and dst_bias is not None:
            assert dst_bias.numel() == nblk * 32 and dst_bias.dtype == torch.float32
            bias = dst_bias
        elif with_bias:
            bias = self.f32(nblk * 32)
        self.packed_tensors.append((w, bias))
        lay = L.PackLayer()
        lay.dst_w = w.data_ptr()
        lay.dst_bias = bias.data_ptr() if bias is not None else None
        lay.src_bias = bias_ptr or None
        lay.cout, lay.cin, lay.nblk, lay.nchunk, lay.taps, lay.perm = cout, cin, nblk, nchunk, taps, perm
        lay.seg_begin, lay.seg_count = len(self.pack_segs), len(segs)
        for s in segs:
            ps = L.PackSeg()
            ps.src = s["src"]
            ps.cin_src, ps.cout_src = s["cin_src"], s["cout_src"]
            ps.c_begin, ps.c_count = s["c_begin"], s["c_count"]
            ps.src_off, ps.src_o_off = s.get("src_off", 0), s.get("src_o_off", 0)
            ps.transposed = s.get("transposed", 0)
            ps.scale = s.get("scale", 1.0)
            self.pack_segs.append(ps)
            self.pack_scale_ptrs.append(s.get("scale_ptr", 0))
        self.pack_layers.append(lay)
        self.pack_max_elems = max(self.pack_max_elems, nel)
        return w.data_ptr(), (bias.data_ptr() if bias is not None else 0)

    # False (SRX_FLAT_PACK=0): the (units x layers) grid of srx_pack_weights_scaled (A/B runs, tests)
    flat_packing = os.environ.get("SRX_FLAT_PACK", "1") != "0"

    def finalize_packing(self):
        if not self.pack_layers:
            return
        la = (L.PackLayer * len(self.pack_layers))(*self.pack_layers)
        sa = (L.PackSeg * len(self.pack_segs))(*self.pack_segs)
        pa = (C.c_void_p * len(self.pack_scale_ptrs))(*[p or None for p in self.pack_scale_ptrs])
        self._pack_dev = (self.upload(la), self.upload(sa), self.upload(pa))
        self._pack_scaled = any(self.pack_scale_ptrs)
        # flat (layer, unit) table for srx_pack_weights_flat, and where each layer's units start in it
        starts, ents = [], []
        for li, lay in enumerate(self.pack_layers):
            starts.append(len(ents))
            n_units = lay.nblk * lay.nchunk
            assert li < 65536 and n_units <= 65536
            ents += [(li << 16) | u for u in range(n_units)]
        starts.append(len(ents))
        self._pack_unit_start = starts
        self._pack_units_dev = self.upload((C.c_uint32 * max(1, len(ents)))(*ents))

    def run_packing(self, first: int = 0, count: Optional[int] = None, stream: Optional[torch.cuda.Stream] = None):
        """Pack layers [first, first+count) of the table (default: all) on `stream` (default: current)."""
        if self._pack_dev is None:
            return
        count = len(self.pack_layers) - first if count is None else count
        if count <= 0:
            return
        la, sa, pa = self._pack_dev
        sp = stream.cuda_stream if stream is not None else stream_ptr(self.device)
        u0, u1 = self._pack_unit_start[first], self._pack_unit_start[first + count]
        if self.flat_packing and u1 > u0:
            rc = self.lib.srx_pack_weights_flat(self.dt, la.data_ptr(), sa.data_ptr(),
                                                pa.data_ptr() if self._pack_scaled else None,
                                                self._pack_units_dev.data_ptr() + 4 * u0, u1 - u0, sp)
        else:
            rc = self.lib.srx_pack_weights_scaled(self.dt, la.data_ptr() + first * C.sizeof(L.PackLayer), sa.data_ptr(),
                                                  pa.data_ptr() if self._pack_scaled else None,
                                                  count, self.pack_max_elems, sp)
        L.check(rc, "srx_pack_weights")

    # --------------------------------------------------------------- ops
    def conv(self, ops: list, x: View, y: View, w_ptr: int, bias_ptr: int, B: int, H: int, W: int,
             cin: int, cout: int, *, ksize=3, stride=1, act=1.0, alpha=1.0, ps=0,
             mask: Optional[View] = None, mask_slope=0.0,
             res1: Optional[View] = None, beta1=1.0, res2: Optional[View] = None, beta2=1.0):
        op = L.Op()
        op.kind = L.OP_CONV
        d = op.u.conv
        d.dtype, d.ksize, d.stride = self.dt, ksize, stride
        d.B, d.H, d.W, d.cin, d.cout, d.pixel_shuffle = B, H, W, cin, cout, ps
        d.x, d.x_img_stride = x[0].ptr(x[1]), x[0].img_stride
        d.y, d.y_img_stride = y[0].ptr(y[1]), y[0].img_stride
        d.w, d.bias = w_ptr, (bias_ptr or None)
        d.act_slope, d.alpha = act, alpha
        if mask is not None:
            d.mask, d.mask_img_stride, d.mask_slope = mask[0].ptr(mask[1]), mask[0].img_stride, mask_slope
        if res1 is not None:
            d.res1, d.res1_img_stride, d.beta1 = res1[0].ptr(res1[1]), res1[0].img_stride, beta1
        if res2 is not None:
            d.res2, d.res2_img_stride, d.beta2 = res2[0].ptr(res2[1]), res2[0].img_stride, beta2
        ops.append(op)
        if isinstance(ops, OpList):
            pad = ksize // 2
            OH, OW = (H + 2 * pad - ksize) // stride + 1, (W + 2 * pad - ksize) // stride + 1
            ops.meta.append((conv_kernel_name(self.dt, ksize, stride, OH, OW, cout, B),
                             2.0 * B * OH * OW * ksize * ksize * cin * cout))

    def generic(self, ops: list, kind: int, *, a: Optional[View] = None, b: Optional[View] = None,
                c: Optional[View] = None, c_ptr: int = 0, mask: Optional[View] = None,
                B=0, C_=0, H=0, W=0, noct=0, f0=0.0, f1=0.0, n=0):
        op = L.Op()
        op.kind = kind
        g = op.u.g
        g.dtype, g.B, g.C, g.H, g.W, g.noct = self.dt, B, C_, H, W, noct
        if a is not None:
            g.a, g.a_img_stride = a[0].ptr(a[1]), a[0].img_stride
        if b is not None:
            g.b, g.b_img_stride = b[0].ptr(b[1]), b[0].img_stride
        if c is not None:
            g.c, g.c_img_stride = c[0].ptr(c[1]), c[0].img_stride
        elif c_ptr:
            g.c = c_ptr
        if mask is not None:
            g.mask, g.mask_img_stride = mask[0].ptr(mask[1]), mask[0].img_stride
        g.f0, g.f1, g.n = f0, f1, n
        ops.append(op)
        if isinstance(ops, OpList):
            ops.meta.append(("op%d" % kind, 0.0))

    def memset(self, ops: list, t: torch.Tensor):
        self.generic(ops, L.OP_MEMSET, c_ptr=t.data_ptr(), n=t.numel() * t.element_size())

    # ------------------------------------------------------------- wgrad
    @staticmethod
    def wgrad_tasks_for_layer(cin: int, cout: int, x_oct0: int, gy_oct0: int, out_ptr: int,
                              scale: float = 1.0, bias_ptr: int = 0, x_ptr: int = 0, gy_ptr: int = 0) -> List[dict]:
        """64x64 block tasks of one convolution's weight gradient.
        out_ptr: tap-major fp32 [taps][cout][cin] tensor; bias_ptr: fp32 [cout] or 0;
        x_ptr / gy_ptr: image-0 base pointers of the views (0: the launch descriptor's)."""
        tasks: Dict[Tuple[int, int], dict] = {}
        aligned = gy_oct0 % 4 == 0       # 32-channel blocks of the view line up with the 64-wide tasks
        for co32 in range(_ceil(cout, 32)):
            if aligned:
                blk = gy_oct0 // 4 + co32
                gy_oct, b = (blk // 2) * 8, blk % 2
            else:
                gy_oct, b = gy_oct0 + (co32 // 2) * 8, co32 % 2
            co0 = co32 * 32
            noct_g = 4 * b + min(4, _ceil(cout, 8) - co32 * 4)
            for ci in range(_ceil(cin, 64)):
                t = tasks.setdefault((ci, gy_oct), {
                    "x": x_ptr, "gy": gy_ptr, "x_oct": x_oct0 + ci * 8, "gy_oct": gy_oct,
                    "x_noct": min(8, _ceil(cin, 8) - ci * 8), "gy_noct": 0, "sub": {}, "bias": {}})
                t["gy_noct"] = max(t["gy_noct"], noct_g)
                if ci == 0 and bias_ptr:
                    t["bias"][b] = dict(out=bias_ptr, co0=co0, n=cout, scale=scale)
                for a in (0, 1):
                    ci0 = ci * 64 + a * 32
                    if ci0 < cin:
                        t["sub"][a + 2 * b] = dict(out=out_ptr, ld=cin, co_n=cout, ci0=ci0, co0=co0, scale=scale)
        return list(tasks.values())

    @staticmethod
    def merge_tasks(tasks: List[dict]) -> List[dict]:
        """Tasks that read the same (x block, gy block) pair share one workgroup as long as
        their 32x32 sub-block slots (and bias slots) do not collide."""
        merged: Dict[Tuple[int, int, int, int], List[dict]] = {}
        for t in tasks:
            lst = merged.setdefault((t["x"], t["gy"], t["x_oct"], t["gy_oct"]), [])
            for m in lst:
                if not (set(m["sub"]) & set(t["sub"])) and not (set(m["bias"]) & set(t["bias"])):
                    m["sub"].update(t["sub"])
                    m["bias"].update(t["bias"])
                    m["x_noct"] = max(m["x_noct"], t["x_noct"])
                    m["gy_noct"] = max(m["gy_noct"], t["gy_noct"])
                    break
            else:
                lst.append({**t, "sub": dict(t["sub"]), "bias": dict(t["bias"])})
        return [m for lst in merged.values() for m in lst]

    # Off: with seven tasks per dense block instead of eight, the XCD-affine block order (all tasks of
    # a block on one XCD, sharing its L2) packs 4 x 7 = 28 workgroups on an XCD's 32 CUs where it
    # packed 4 x 8 = 32 — an eighth less work on an eighth fewer CUs, same time (measured: one
    # pixel slice more per task overflows the XCD instead: 255 -> 403 us per launch).
    pair_wgrad_tasks = False
    MAX_TASK_SKIP = 24      # octet planes a task's two groups may lie apart (32 planes span in all)

    @staticmethod
    def pair_sparse_tasks(tasks: List[dict]) -> List[dict]:
        """Two tasks that each keep ONE 32x32 sub-block busy (the odd 32-channel slices of a dense
        block: y1 x gy2 of conv2, y3 x gy4 of conv4) become one task whose first four x / gy
        octets are the first task's group and whose last four are the second's (x_skip / gy_skip
        planes apart): sub-blocks (0,0) and (1,1) are used, the cross terms stay empty.  One
        workgroup pass over the pixels less per pair."""
        def single(t):
            return len(t["sub"]) == 1 and not t["bias"] and not t.get("x_skip") and not t.get("gy_skip")

        out, pool = [], []
        for t in tasks:
            (pool if single(t) else out).append(t)
        pool.sort(key=lambda t: (t["x"], t["gy"], t["x_oct"] + 4 * (next(iter(t["sub"])) & 1)))
        while pool:
            t1 = pool.pop(0)
            s1 = next(iter(t1["sub"]))
            xg1, gg1 = t1["x_oct"] + 4 * (s1 & 1), t1["gy_oct"] + 4 * (s1 >> 1)
            mate = None
            for t2 in pool:
                s2 = next(iter(t2["sub"]))
                xg2, gg2 = t2["x_oct"] + 4 * (s2 & 1), t2["gy_oct"] + 4 * (s2 >> 1)
                if (t2["x"], t2["gy"]) == (t1["x"], t1["gy"]) and 4 <= xg2 - xg1 <= 4 + Plan.MAX_TASK_SKIP \
                        and 4 <= gg2 - gg1 <= 4 + Plan.MAX_TASK_SKIP:
                    mate = (t2, s2, xg2, gg2)
                    break
            full1 = min(4, t1["x_noct"] - 4 * (s1 & 1)) == 4 and min(4, t1["gy_noct"] - 4 * (s1 >> 1)) == 4
            if mate is None or not full1:
                out.append(t1)
                continue
            t2, s2, xg2, gg2 = mate
            pool.remove(t2)
            out.append({"x": t1["x"], "gy": t1["gy"], "x_oct": xg1, "gy_oct": gg1,
                        "x_noct": 4 + min(4, t2["x_noct"] - 4 * (s2 & 1)),
                        "gy_noct": 4 + min(4, t2["gy_noct"] - 4 * (s2 >> 1)),
                        "x_skip": xg2 - xg1 - 4, "gy_skip": gg2 - gg1 - 4,
                        "sub": {0: t1["sub"][s1], 3: t2["sub"][s2]}, "bias": {}})
        return out

    def wgrad(self, ops: list, x: Optional[View], gy: Optional[View], tasks: List[dict], B: int, H: int, W: int, *,
              ksize=3, stride=1, target_blocks=512, nslices: Optional[int] = None,
              x_img_stride: int = 0, gy_img_stride: int = 0, group: int = 0, tile_walk: bool = False):
        arr = (L.WgradTask * len(tasks))()
        for i, t in enumerate(tasks):
            k = arr[i]
            k.x, k.gy = (t.get("x") or None), (t.get("gy") or None)
            k.x_oct, k.gy_oct, k.x_noct, k.gy_noct = t["x_oct"], t["gy_oct"], t["x_noct"], t["gy_noct"]
            k.x_skip, k.gy_skip = t.get("x_skip", 0), t.get("gy_skip", 0)
            for s, e in t["sub"].items():
                k.out[s] = e["out"]
                k.ld[s], k.co_n[s], k.ci0[s], k.co0[s], k.scale[s] = e["ld"], e["co_n"], e["ci0"], e["co0"], e["scale"]
            for b, e in t.get("bias", {}).items():
                k.bias_out[b] = e["out"]
                k.bias_co0[b], k.bias_n[b], k.bias_scale[b] = e["co0"], e["n"], e["scale"]
        dev = self.upload(arr)
        op = L.Op()
        op.kind = L.OP_WGRAD
        d = op.u.wgrad
        d.dtype, d.ksize, d.stride, d.B, d.H, d.W = self.dt, ksize, stride, B, H, W
        if x is not None:
            d.x, d.x_img_stride = x[0].ptr(x[1]), x[0].img_stride
        else:
            d.x_img_stride = x_img_stride
        if gy is not None:
            d.gy, d.gy_img_stride = gy[0].ptr(gy[1]), gy[0].img_stride
        else:
            d.gy_img_stride = gy_img_stride
        d.tasks, d.ntasks = dev.data_ptr(), len(tasks)
        d.nslices = nslices if nslices is not None else max(1, _ceil(target_blocks, len(tasks)))
        d.group = group
        pad = ksize // 2
        OH, OW = (H + 2 * pad - ksize) // stride + 1, (W + 2 * pad - ksize) // stride + 1
        tile_walk = bool(tile_walk or self.wgrad_walk) and ksize == 3 and stride == 1 and OW >= 24
        d.tile_walk = 1 if tile_walk else 0
        if self.deterministic_wgrad:
            assert self._stage_total > 0, "deterministic weight gradients need the plan's staging buffer"
            d.nslices = min(d.nslices, self.det_slices)
            d.slice_stride = self._stage_total
            # the unpack launch adds up as many copies of a tensor as the launches that fill it use
            for t in tasks:
                for e in list(t["sub"].values()) + list(t.get("bias", {}).values()):
                    it = self._item_by_src.get(e["out"])
                    if it is not None:
                        it.nslices = max(it.nslices, d.nslices)
        ops.append(op)
        if isinstance(ops, OpList):
            pairs = sum(min(32, e["ld"] - e["ci0"]) * min(32, e["co_n"] - e["co0"])
                        for t in tasks for e in t["sub"].values())
            ltw = 5 if OW >= 24 else (4 if OW >= 12 else 3)
            f = self.dt == L.F32
            tp = 128 if ksize == 1 else ((64 if f else 128) if stride == 1 else (32 if f else 64))
            name = "wgrad_kernel<%s,%d,%d,%d,%d>" % ("float" if f else "bf16", stride,
                                                     5 if ksize == 1 else ltw, ksize, tp)
            if tile_walk:
                name = "wgrad_walk_kernel<%s,%d>" % ("float" if f else "bf16", tp)
            ops.meta.append((name, 2.0 * B * OH * OW * ksize * ksize * pairs))

    # The column-walk weight-gradient kernel (3x3, stride 1, images at least 24 pixels wide) for every
    # launch it serves: a third fewer staged activation bytes per tile at the same register budget as
    # the plain kernel, so it shares a CU with the data-gradient chain just as well.  Config 2, one box,
    # per step: 9.43 / 9.46 ms with the walk in the last two trunk launches only, 9.28 / 9.33 in every
    # trunk launch, 9.22 / 9.22 in every launch.  SRX_WGRAD_WALK=0: only where a launch asks for it.
    wgrad_walk = os.environ.get("SRX_WGRAD_WALK", "1") != "0"

    unpack_accumulate = 0    # 1: run_unpack adds into the destination gradients

    # Deterministic weight gradients (opt-in): by default the pixel slices of a weight-gradient task
    # add into one staging tensor with fp32 atomics, whose order — and so the last bits of the
    # gradients — varies between runs.  With this set (before the plan is built), every slice
    # stores into its own copy of the staging buffer and the unpack launch adds the copies in
    # slice order: same inputs, same bits.  At most det_slices slices per task are then used
    # (det_slices copies of the gradient-sized staging buffer).
    deterministic_wgrad = False
    det_slices = 32
    _stage_total = 0

    def unpack_item(self, src_ptr: int, dst_ptr: int, cout: int, cin: int, taps: int):
        it = L.UnpackItem()
        it.src, it.dst, it.cout, it.cin, it.taps = src_ptr, dst_ptr, cout, cin, taps
        it.accumulate = self.unpack_accumulate
        if self.deterministic_wgrad and self._stage_total > 0:
            it.nslices, it.slice_stride = 1, self._stage_total
            self._item_by_src[src_ptr] = it
        self.unpack_items.append(it)
        self.unpack_max = max(self.unpack_max, cout * cin * taps)

    def trim_stage_memset(self, ops: list) -> None:
        """Deterministic form: the staging buffer holds det_slices copies of the gradient tensors, of
        which the launches of this plan use only the first max(nslices); the recorded memset of the
        buffer (ops built so far, not yet frozen) is cut to those copies."""
        if not (self.deterministic_wgrad and self._stage_total > 0):
            return
        used = max([it.nslices for it in self.unpack_items] + [1])
        for op in ops:
            if op.kind == L.OP_MEMSET and op.u.g.c == self.stage.data_ptr():
                op.u.g.n = min(op.u.g.n, used * self._stage_total * 4)

    def finalize_unpack(self):
        if self.unpack_items:
            arr = (L.UnpackItem * len(self.unpack_items))(*self.unpack_items)
            self._unpack_dev = self.upload(arr)

    def run_unpack(self, first: int = 0, count: Optional[int] = None):
        """Unpack table entries [first, first+count) (default: all) on the current stream."""
        if self._unpack_dev is not None:
            count = len(self.unpack_items) - first if count is None else count
            L.check(self.lib.srx_unpack_wgrad(self._unpack_dev.data_ptr() + first * C.sizeof(L.UnpackItem), count,
                                              self.unpack_max, stream_ptr(self.device)), "srx_unpack_wgrad")

    # --------------------------------------------------------------- run
    @staticmethod
    def freeze(ops: list):
        return (L.Op * len(ops))(*ops) if ops else None

    def run(self, frozen, stream: Optional[torch.cuda.Stream] = None) -> None:
        if frozen is None:
            return
        sp = stream.cuda_stream if stream is not None else stream_ptr(self.device)
        n = len(frozen)
        L.check(self.lib.srx_run_ops(C.addressof(frozen), n, sp), "srx_run_ops")

    # ------------------------------------------------- two-stream op lists
    # Weight-gradient launches depend only on tensors the data-gradient chain has already
    # produced, and nothing in that chain waits for them.  A split op list keeps the chain on the
    # caller's stream and moves each weight-gradient launch to a side stream behind an event, so
    # its workgroups share the CUs with the chain's (one wave per SIMD each) and fill the MFMA
    # issue slots the chain's short launches leave idle.
    @staticmethod
    def split_ops(main: "OpList", side_marks: List[Tuple[int, "OpList", List[str]]]):
        """main: ops of the chain; side_marks: (number of chain ops that must precede, ops,
        names of the parameters whose gradients those ops complete).
        Returns (segments, serial ops, serial meta, names per segment, meta per segment):
        segments = [(chain ops, side ops|None)]; the serial list interleaves both in dependency
        order for one-stream runs; meta per segment = (chain meta, side meta) for profiling the
        two-stream form."""
        segs, serial, meta, seg_meta = [], [], [], []
        pos = 0
        for after, ops, names in side_marks:
            if after > pos or not segs:
                chunk = main[pos:after]
                segs.append([list(chunk), list(ops), list(names)])
                seg_meta.append([list(main.meta[pos:after]), list(ops.meta)])
                serial += chunk
                meta += main.meta[pos:after]
                pos = after
            else:
                segs[-1][1] += list(ops)
                segs[-1][2] += list(names)
                seg_meta[-1][1] += list(ops.meta)
            serial += ops
            meta += ops.meta
        if pos < len(main) or not segs:
            segs.append([list(main[pos:]), [], []])
            seg_meta.append([list(main.meta[pos:]), []])
            serial += main[pos:]
            meta += main.meta[pos:]
        frozen = [(Plan.freeze(a), Plan.freeze(b)) for a, b, _ in segs]
        return frozen, Plan.freeze(serial), meta, [n for _, _, n in segs], seg_meta

    def run_split(self, segments, side_hook=None) -> None:
        """side_hook(i), if given, is called with the side stream current right after segment
        i's side ops were enqueued (gradient exchange of what they completed)."""
        main = torch.cuda.current_stream(self.device)
        if getattr(self, "_side_stream", None) is None:
            self._side_stream = side_stream(self.device)
            self._side_events = []
            self._side_done = torch.cuda.Event()
        side = self._side_stream
        used = False
        for i, (chain, aside) in enumerate(segments):
            self.run(chain)
            if aside is not None:
                if i >= len(self._side_events):
                    self._side_events.append(torch.cuda.Event())
                ev = self._side_events[i]
                ev.record(main)
                side.wait_event(ev)
                self.run(aside, stream=side)
                if side_hook is not None:
                    with torch.cuda.stream(side):
                        side_hook(i)
                used = True
        if used:
            self._side_done.record(side)
            main.wait_event(self._side_done)

    # ----------------------------------------------------- layout helpers
    def nchw_to_c8(self, src: torch.Tensor, dst: View):
        buf, o0 = dst
        B, C_, H, W = src.shape
        if (B, H, W) != (buf.B, buf.H, buf.W) or _ceil(C_, 8) + o0 > buf.C8:
            raise ValueError(f"tensor of shape {tuple(src.shape)} does not fit the plan's buffer")
        src = src.detach().contiguous().float()
        self._last_src = src        # keep alive until the stream has consumed it
        L.check(self.lib.srx_nchw_to_c8(self.dt, src.data_ptr(), buf.ptr(o0), buf.img_stride, B, C_, H, W,
                                        stream_ptr(self.device)), "srx_nchw_to_c8")

    def c8_to_nchw(self, src: View, C_: int) -> torch.Tensor:
        buf, o0 = src
        out = torch.empty((buf.B, C_, buf.H, buf.W), dtype=torch.float32, device=self.device)
        L.check(self.lib.srx_c8_to_nchw(self.dt, buf.ptr(o0), buf.img_stride, out.data_ptr(), buf.B, C_,
                                        buf.H, buf.W, stream_ptr(self.device)), "srx_c8_to_nchw")
        return out


def conv_seg(weight: torch.Tensor, scale: float = 1.0, scale_ptr: int = 0) -> dict:
    """Single forward segment for a [cout][cin][k][k] fp32 weight."""
    return dict(src=weight.data_ptr(), cin_src=weight.shape[1], cout_src=weight.shape[0],
                c_begin=0, c_count=weight.shape[1], scale=scale, scale_ptr=scale_ptr)


def dgrad_seg(weight: torch.Tensor, c_begin: int, src_cin_off: int, scale: float = 1.0,
              scale_ptr: int = 0) -> dict:
    """Transposed segment: virtual input channels [c_begin, c_begin+cout_src) are this
    convolution's output channels; virtual output channel n is its input channel
    src_cin_off + n."""
    return dict(src=weight.data_ptr(), cin_src=weight.shape[1], cout_src=weight.shape[0],
                c_begin=c_begin, c_count=weight.shape[0], src_off=src_cin_off, src_o_off=0,
                transposed=1, scale=scale, scale_ptr=scale_ptr)


# ===========================================================================
# Generator (esrgan/generator.py:50-72) and its blocks
# ===========================================================================
class _GenBase(Plan):
    """Emitters shared by the whole-generator plan and the stand-alone block plans."""

    def setup_grads(self, P: Dict[str, torch.Tensor], flat_grad: Optional[torch.Tensor] = None):
        """Flat fp32 gradient buffer in parameter order (+ per-name views), the tap-major
        staging buffer the kernels accumulate into, and the unpack table between them."""
        self.param_names = [n for n in P]
        total = sum(P[n].numel() for n in self.param_names)
        self.flat_grad = flat_grad if flat_grad is not None else self.f32(total, zero=True)
        assert self.flat_grad.numel() >= total
        self.grads: Dict[str, torch.Tensor] = {}
        self._stage_total = total
        self.stage = self.f32(total * (self.det_slices if self.deterministic_wgrad else 1), zero=True)
        self._soff: Dict[str, int] = {}
        off = 0
        for n in self.param_names:
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

    def fwd_layer(self, P, name, perm=0, dst_w=None, dst_bias=None):
        wt = P[name + ".weight"]
        return self.packed_layer(wt.shape[0], wt.shape[1], 9, perm, [conv_seg(wt)],
                                 P[name + ".bias"].data_ptr(), dst_w=dst_w, dst_bias=dst_bias)

    # One launch per dense block forward (csrc/rdb.hip) where the shape allows it (bf16, 64 block
    # channels, 32 growth channels) and where it is the faster form: while its 16 x 8-pixel tiles
    # fit the chip in at most two rounds (one workgroup per CU).  Per training step, A/B on one box,
    # fused against five launches per block: 16 x 32 x 32 (128 tiles, config 3) 9.27 against 9.9 ms;
    # 16 x 64 x 64 (512 tiles, config 2) 10.13 against 10.47; 32 x 64 x 64 (1 024 tiles) 18.1 against
    # 17.1; 16 x 128 x 128 34.7 against 30.6.
    # SRX_FUSED_RDB=0 / 1 forces the five-launch / the fused form (A/B runs, tests).
    fused_rdb = {"0": False, "1": True}.get(os.environ.get("SRX_FUSED_RDB", "auto"))
    fused_rdb_max_tiles = 512

    def use_fused_rdb(self, B: int, h: int, w: int, nf: int, gc: int) -> bool:
        if self.dt != L.BF16 or (nf, gc) != (64, 32) or self.fused_rdb is False:
            return False
        return bool(self.fused_rdb) or B * _ceil(h, 8) * _ceil(w, 16) <= self.fused_rdb_max_tiles

    # The same for the data gradient of a dense block (srx_rdb_backward).  Beside it the weight
    # gradients of the side stream find no room on a CU it occupies (132 KB of LDS): only where its
    # tiles leave CUs free (config 3, 128 tiles: 8.99 -> 8.41 ms per step; config 2, 512 tiles: 10.1 ->
    # 11.9).  SRX_FUSED_RDB_BWD=0 / 1 forces the choice.
    fused_rdb_bwd = {"0": False, "1": True}.get(os.environ.get("SRX_FUSED_RDB_BWD", "auto"))
    fused_rdb_bwd_max_tiles = 256

    def use_fused_rdb_bwd(self, B: int, h: int, w: int, nf: int, gc: int) -> bool:
        if self.dt != L.BF16 or (nf, gc) != (64, 32) or self.fused_rdb_bwd is False:
            return False
        return bool(self.fused_rdb_bwd) or B * _ceil(h, 8) * _ceil(w, 16) <= self.fused_rdb_bwd_max_tiles

    # Tile of the fused kernels: 16 x 8 pixels per workgroup; 8 x 8 for inference plans whose feature
    # map that still fits the chip in one round (twice the workgroups for a fifth more arithmetic:
    # 16 x 32 x 32 stand-alone 23.5 -> 20.7 us per block).  Training plans keep 16 x 8: the work of
    # the side stream (weight gradients in backward; the discriminator / VGG passes on the real batch
    # beside the generator's forward in the adversarial step) runs on the CUs the larger tiles leave
    # free (config 3, A/B on one box: 7.84 ms per step with 16 x 8 in both passes, 8.00 with 8 x 8 in
    # forward, 8.12 in both).  SRX_RDB_TILE_FWD / SRX_RDB_TILE_BWD = 0 / 1 / 2 force 16 x 8 / 8 x 8 / two stacked 16 x 8.
    rdb_tile_fwd = {"0": 0, "1": 1, "2": 2}.get(os.environ.get("SRX_RDB_TILE_FWD", "auto"))
    rdb_tile_bwd = {"0": 0, "1": 1, "2": 2}.get(os.environ.get("SRX_RDB_TILE_BWD", "auto"))

    def rdb_tile(self, B: int, h: int, w: int, backward: bool) -> int:
        """srx_rdb_desc.tile: 0 = 16 x 8 pixels per workgroup, 1 = 8 x 8, 2 = two 16 x 8 tiles stacked
        vertically per workgroup (one staged input region for both)."""
        forced = self.rdb_tile_bwd if backward else self.rdb_tile_fwd
        if forced is not None:
            return forced
        if B * _ceil(h, 8) * _ceil(w, 16) > 256:
            return 2            # more 16 x 8 tiles than CUs: one round of two-tile workgroups instead of two rounds
        if backward or self.need_grad:
            return 0
        return 1 if B * _ceil(h, 8) * _ceil(w, 8) <= 256 else 0

    def rdb_fwd(self, ops: list, cur: Buf, dst: View, w_ptr: int, bias_ptr: int, B: int, h: int, w: int,
                nf: int, gc: int, store_growth: bool, alpha: float, beta1: float,
                res2: Optional[View] = None, beta2: float = 0.0, backward_of: Optional[Buf] = None):
        """One-launch dense block: forward over the block buffer `cur`, or (backward_of = the block's
        activation buffer) its data gradient over the gradient buffer `cur`."""
        op = L.Op()
        op.kind = L.OP_RDB_FWD if backward_of is None else L.OP_RDB_BWD
        d = op.u.rdb
        d.dtype, d.B, d.H, d.W, d.nf, d.gc = self.dt, B, h, w, nf, gc
        d.store_growth = 1 if store_growth else 0
        d.tile = self.rdb_tile(B, h, w, backward_of is not None)
        d.buf, d.buf_img_stride = cur.ptr(0), cur.img_stride
        d.y, d.y_img_stride = dst[0].ptr(dst[1]), dst[0].img_stride
        d.w, d.bias = w_ptr, bias_ptr or None
        d.act_slope, d.alpha, d.beta1, d.beta2 = LRELU, alpha, beta1, beta2
        if res2 is not None:
            d.res2, d.res2_img_stride = res2[0].ptr(res2[1]), res2[0].img_stride
        if backward_of is not None:
            d.act, d.act_img_stride = backward_of.ptr(0), backward_of.img_stride
        ops.append(op)
        if isinstance(ops, OpList):
            macs = sum(9 * (nf + k * gc) * gc for k in range(4)) + 9 * (nf + 4 * gc) * nf
            ops.meta.append(("rdb_fwd_kernel" if backward_of is None else "rdb_bwd_kernel", 2.0 * B * h * w * macs))

    def dgrad_layer(self, P, name, cout_v):
        wt = P[name + ".weight"]
        return self.packed_layer(cout_v, wt.shape[0], 9, 0, [dgrad_seg(wt, 0, 0)], with_bias=False)[0]

    # Workgroups (tasks x pixel slices) of the weight-gradient launch of a single layer (the layers
    # outside the dense blocks; in backward they run beside the data gradients of the HR-resolution
    # layers, whose workgroups fill every CU twice over): few enough to leave those room.  A/B on one
    # box, per step: config 2 9.56 ms at 512, 9.50 at 256, 9.08-9.16 at 128-160, 9.90 at 64; config 3 7.04
    # at 512, 6.78 at 128; 64 x 128 x 128 (16 M pixels at HR resolution) 119.8 at 512, 121.7 at 128.
    # SRX_LAYER_WGRAD_BLOCKS overrides both.
    layer_wgrad_blocks_small, layer_wgrad_blocks_large = 160, 512
    layer_wgrad_small_pixels = 1 << 21
    _layer_wgrad_blocks_env = os.environ.get("SRX_LAYER_WGRAD_BLOCKS")

    def layer_wgrad(self, ops, P, name, x: View, gy: View, B, Hh, Ww):
        wt = P[name + ".weight"]
        tasks = self.wgrad_tasks_for_layer(wt.shape[1], wt.shape[0], 0, 0, self.sp(name + ".weight"),
                                           bias_ptr=self.sp(name + ".bias"))
        blocks = (int(self._layer_wgrad_blocks_env) if self._layer_wgrad_blocks_env else
                  self.layer_wgrad_blocks_small if B * Hh * Ww <= self.layer_wgrad_small_pixels
                  else self.layer_wgrad_blocks_large)
        self.wgrad(ops, x, gy, tasks, B, Hh, Ww, target_blocks=blocks)

    # ---- dense block ------------------------------------------------------
    def emit_rdb_fwd(self, ops, P, pre, cur: Buf, dst: View, B, h, w, nf, gc, q,
                     rrdb_in: Optional[View]):
        """ResidualDenseBlock.forward (esrgan/generator.py:17-23); q == 2 also applies the
        enclosing RRDB's ``out*0.2 + x`` (:37) in the same epilogue."""
        nf8, gc8 = nf // 8, gc // 8
        if self.use_fused_rdb(B, h, w, nf, gc):
            # the five layers packed back to back: conv1..conv4 (4, 6, 8, 10 chunks of 9 x 512), conv5 (2 x 12)
            chunks = [(nf + k * gc) // 16 for k in range(4)] + [2 * (nf + 4 * gc) // 16]
            w_all = torch.empty(sum(chunks) * 9 * 512, dtype=self.tdtype, device=self.device)
            b_all = self.f32(4 * 32 + 64)
            self.keep.append(w_all)
            self.bytes_alloc += w_all.numel() * self.esize
            wo = bo = 0
            for k in range(1, 6):
                nw, nb_ = chunks[k - 1] * 9 * 512, (32 if k < 5 else 64)
                self.fwd_layer(P, pre + f"conv{k}", dst_w=w_all[wo:wo + nw], dst_bias=b_all[bo:bo + nb_])
                wo, bo = wo + nw, bo + nb_
            if q < 2:
                self.rdb_fwd(ops, cur, dst, w_all.data_ptr(), b_all.data_ptr(), B, h, w, nf, gc, self.need_grad,
                             RES, 1.0)
            else:
                self.rdb_fwd(ops, cur, dst, w_all.data_ptr(), b_all.data_ptr(), B, h, w, nf, gc, self.need_grad,
                             RES * RES, RES, rrdb_in, 1.0)
            return
        packed = {k: self.fwd_layer(P, pre + f"conv{k}") for k in range(1, 5)}
        for k in range(1, 5):
            self.conv(ops, (cur, 0), (cur, nf8 + (k - 1) * gc8), packed[k][0], packed[k][1], B, h, w,
                      nf + (k - 1) * gc, gc, act=LRELU)
        wp, bp = self.fwd_layer(P, pre + "conv5")
        if q < 2:
            self.conv(ops, (cur, 0), dst, wp, bp, B, h, w, nf + 4 * gc, nf,
                      alpha=RES, res1=(cur, 0), beta1=1.0)
        else:
            self.conv(ops, (cur, 0), dst, wp, bp, B, h, w, nf + 4 * gc, nf,
                      alpha=RES * RES, res1=(cur, 0), beta1=RES, res2=rrdb_in, beta2=1.0)

    def emit_rdb_bwd(self, ops, P, pre, cur: Buf, G: Buf, dst: View, B, h, w, nf, gc, q,
                     rrdb_g: Optional[View], defer_wgrad: Optional[list] = None):
        """Backward of one dense block.  G = [gy1|gy2|gy3|gy4|g_out]; g_out (the gradient
        of the block output, or of the RRDB output when q == 2) must already be in place.
        dst receives the gradient of the block input (+ the RRDB skip gradient rrdb_g when
        q == 0)."""
        nf8, gc8 = nf // 8, gc // 8
        g8 = 4 * gc8
        s5 = RES * RES if q == 2 else RES
        W = [P[pre + f"conv{k}.weight"] for k in range(1, 6)]
        dg = {}
        for j in range(4, -1, -1):
            segs, c0 = [], 0
            src_off = 0 if j == 0 else nf + (j - 1) * gc
            for k in range(j + 1, 6):
                segs.append(dgrad_seg(W[k - 1], c0, src_off, s5 if k == 5 else 1.0))
                c0 += W[k - 1].shape[0]
            dg[j] = (segs, c0)
        fused = self.use_fused_rdb_bwd(B, h, w, nf, gc)
        if fused:
            # the five data-gradient layers packed back to back, j = 4 .. 0 (4, 6, 8, 10, 2 x 12 chunks)
            chunks = [_ceil(dg[j][1], 16) * (1 if j else nf // 32) for j in range(4, -1, -1)]
            w_all = torch.empty(sum(chunks) * 9 * 512, dtype=self.tdtype, device=self.device)
            self.keep.append(w_all)
            self.bytes_alloc += w_all.numel() * self.esize
            wo = 0
            for j, nch in zip(range(4, -1, -1), chunks):
                self.packed_layer(nf if j == 0 else gc, dg[j][1], 9, 0, dg[j][0], with_bias=False,
                                  dst_w=w_all[wo:wo + nch * 9 * 512])
                wo += nch * 9 * 512
            if q == 0 and rrdb_g is not None:
                self.rdb_fwd(ops, G, dst, w_all.data_ptr(), 0, B, h, w, nf, gc, True, 1.0, 1.0, rrdb_g, 1.0,
                             backward_of=cur)
            else:
                self.rdb_fwd(ops, G, dst, w_all.data_ptr(), 0, B, h, w, nf, gc, True, 1.0,
                             RES if q == 2 else 1.0, backward_of=cur)
        else:
            for j in dg:
                dg[j] = (self.packed_layer(nf if j == 0 else gc, dg[j][1], 9, 0, dg[j][0], with_bias=False)[0], dg[j][1])
        for j in range(4, -1, -1) if not fused else ():
            wp, c0 = dg[j]
            xin = (G, j * gc8)
            if j > 0:
                self.conv(ops, xin, (G, (j - 1) * gc8), wp, 0, B, h, w, c0, gc,
                          mask=(cur, nf8 + (j - 1) * gc8), mask_slope=LRELU)
            elif q == 0 and rrdb_g is not None:
                self.conv(ops, xin, dst, wp, 0, B, h, w, c0, nf, res1=(G, g8), beta1=1.0,
                          res2=rrdb_g, beta2=1.0)
            else:
                self.conv(ops, xin, dst, wp, 0, B, h, w, c0, nf, res1=(G, g8),
                          beta1=(RES if q == 2 else 1.0))
        # weight + bias gradients of the five convolutions: one launch per block, or deferred
        # into a single launch for the whole trunk (tasks then carry their own buffer pointers)
        tasks = []
        xp, gp = (cur.ptr(0), G.ptr(0)) if defer_wgrad is not None else (0, 0)
        for k in range(1, 6):
            tasks += self.wgrad_tasks_for_layer(nf + (k - 1) * gc, gc if k < 5 else nf, 0, (k - 1) * gc8,
                                                self.sp(pre + f"conv{k}.weight"), s5 if k == 5 else 1.0,
                                                bias_ptr=self.sp(pre + f"conv{k}.bias"), x_ptr=xp, gy_ptr=gp)
        tasks = self.pair_sparse_tasks(self.merge_tasks(tasks)) if self.pair_wgrad_tasks else self.merge_tasks(tasks)
        if defer_wgrad is not None:
            defer_wgrad.extend(tasks)
        else:
            self.wgrad(ops, (cur, 0), (G, 0), tasks, B, h, w)

    # ---- shared run API ---------------------------------------------------
    bwd_segments = None     # two-stream form of bwd_ops (Plan.split_ops), when the plan built one
    bwd_seg_meta = None     # (kernel name, flops) of every op of those segments

    _pack_evs = None        # (parameters final, backward weights packed) when backward's packing runs aside
    _bwd_pack_pending = False
    _n_fwd_layers = 0
    bwd_seg_ranges = None   # per segment: [(first item, item count, flat offset, flat end)] completed by it

    def run_backward_ops(self, on_grads_ready=None) -> bool:
        """Run the backward op lists.  on_grads_ready(lo, hi), if given and the plan has the
        two-stream form, is called (side stream current) as soon as flat_grad[lo:hi] is final
        and unpacked — the hook a data-parallel exchange overlaps with the rest of backward.
        Returns True if the gradients were delivered that way (run_unpack is then not needed)."""
        if self._bwd_pack_pending:
            torch.cuda.current_stream(self.device).wait_event(self._pack_evs[1])
            self._bwd_pack_pending = False
        if self.bwd_segments is None:
            self.run(self.bwd_ops)
            return False
        if on_grads_ready is None or self.bwd_seg_ranges is None:
            self.run_split(self.bwd_segments)
            return False

        def hook(i):
            for first, count, lo, hi in self.bwd_seg_ranges[i]:
                self.run_unpack(first, count)
                on_grads_ready(lo, hi)
        self.run_split(self.bwd_segments, hook)
        return True

    def _ranges_for(self, seg_names: List[List[str]]):
        """Contiguous runs of unpack items / flat-gradient offsets for each segment's names."""
        index = {n: i for i, n in enumerate(self.param_names)}
        sizes = [self.grads[n].numel() for n in self.param_names]
        out, seen = [], set()
        for names in seg_names:
            idx = sorted(index[n] for n in names)
            seen.update(idx)
            runs, k = [], 0
            while k < len(idx):
                j = k
                while j + 1 < len(idx) and idx[j + 1] == idx[j] + 1:
                    j += 1
                lo = self._soff[self.param_names[idx[k]]]
                hi = self._soff[self.param_names[idx[j]]] + sizes[idx[j]]
                runs.append((idx[k], idx[j] - idx[k] + 1, lo, hi))
                k = j + 1
            out.append(runs)
        assert seen == set(range(len(self.param_names))), "a parameter's gradient is completed by no segment"
        return out

    def backward_c8(self) -> None:
        self.run_backward_ops()
        self.run_unpack()


class GeneratorPlan(_GenBase):
    """Forward (+ backward) of GeneratorESRGAN for one input shape.

    params: name -> fp32 CUDA tensor with the reference's state_dict names.
    After backward, ``grads[name]`` are fp32 views of one flat buffer in parameter order.
    """

    trunk_wgrad_slices = 4
    trunk_wgrad_target_wgs = 256   # workgroups a trunk weight-gradient launch aims for (one per CU)
    overlap_wgrad = True        # weight gradients on a side stream, beside the data-gradient chain
    # dense blocks per weight-gradient launch in that mode (0: all in one).  Re-measured with the fused
    # forward in place, A/B on one box, config 2: 4 / 5 / 6 / 8 blocks 9.34 / 9.33 / 9.44 / 9.47 ms per step;
    # 64 x 128 x 128: 122.4 with 5 against 122.9 with 6.
    trunk_wgrad_rdbs = 5
    trunk_wgrad_tail = (4, 3, 2)   # dense blocks of the last launches (they end after the chain: short ones)
    # Beside the fused data gradient (one 132 KB workgroup per tile, which shares a CU with nothing) a
    # weight-gradient launch should leave CUs free: about 130 workgroups instead of one per CU
    # (config 3, one box: 7.57 ms per step with 250 workgroups per launch, 7.08 with 104-156).
    fused_bwd_wgrad_slices = 2
    fused_bwd_wgrad_target_wgs = 128
    # With wgrad_walk off (SRX_WGRAD_WALK=0) the column-walk kernel still serves the last trunk
    # weight-gradient launches, which run after the data-gradient chain has ended, alone on the chip
    # (None: by size — 2 launches at 64 K pixels per step, 8 from 256 K pixels on).  With wgrad_walk on,
    # the default, every eligible launch uses it and this setting changes nothing.
    trunk_wgrad_walk_last = None

    def __init__(self, params: Dict[str, torch.Tensor], B: int, h: int, w: int, tdtype,
                 *, in_ch=3, out_ch=3, nf=64, gc=32, num_blocks=23, need_grad=True,
                 need_input_grad=False, flat_grad: Optional[torch.Tensor] = None,
                 accumulate_grads: bool = False):
        dev = params["conv_first.weight"].device
        super().__init__(dev, tdtype)
        self.unpack_accumulate = 1 if accumulate_grads else 0
        if nf % 8 or gc % 8:
            raise ValueError("num_features and growth channels must be multiples of 8")
        self.B, self.h, self.w = B, h, w
        self.in_ch, self.out_ch, self.nf, self.gc, self.nb = in_ch, out_ch, nf, gc, num_blocks
        self.need_grad, self.need_input_grad = need_grad, need_input_grad
        self.params = P = params
        nR = 3 * num_blocks
        ctot = nf + 4 * gc
        fwd = OpList()
        self.X0 = self.buf(B, in_ch, h, w)
        n_ring = nR + 1 if need_grad else min(nR + 1, 4)
        self.R = [self.buf(B, ctot, h, w) for _ in range(n_ring)]
        Rb = lambda r: self.R[r % n_ring]
        self.F = self.buf(B, nf, h, w) if not need_grad else None   # feat copy when R is a ring
        self.T1 = self.buf(B, nf, h, w)
        self.U0 = self.buf(B, nf, 2 * h, 2 * w)
        self.U1 = self.buf(B, nf, 4 * h, 4 * w)
        self.V = self.buf(B, nf, 4 * h, 4 * w)
        self.Y = self.buf(B, out_ch, 4 * h, 4 * w)

        wp, bp = self.fwd_layer(P, "conv_first")
        self.conv(fwd, (self.X0, 0), (Rb(0), 0), wp, bp, B, h, w, in_ch, nf)
        feat: View = (Rb(0), 0)
        if not need_grad:
            self.conv(fwd, (self.X0, 0), (self.F, 0), wp, bp, B, h, w, in_ch, nf)
            feat = (self.F, 0)
        for r in range(nR):
            blk, q = divmod(r, 3)
            self.emit_rdb_fwd(fwd, P, f"body.{blk}.rdb{q + 1}.", Rb(r), (Rb(r + 1), 0), B, h, w, nf, gc, q,
                              (Rb(r - 2), 0) if q == 2 else None)
        wp, bp = self.fwd_layer(P, "conv_body")
        self.conv(fwd, (Rb(nR), 0), (self.T1, 0), wp, bp, B, h, w, nf, nf, res1=feat, beta1=1.0)
        wp, bp = self.fwd_layer(P, "upsample.0.conv", perm=1)
        self.conv(fwd, (self.T1, 0), (self.U0, 0), wp, bp, B, h, w, nf, 4 * nf, act=LRELU, ps=1)
        wp, bp = self.fwd_layer(P, "upsample.1.conv", perm=1)
        self.conv(fwd, (self.U0, 0), (self.U1, 0), wp, bp, B, 2 * h, 2 * w, nf, 4 * nf, act=LRELU, ps=1)
        wp, bp = self.fwd_layer(P, "conv_hr")
        self.conv(fwd, (self.U1, 0), (self.V, 0), wp, bp, B, 4 * h, 4 * w, nf, nf, act=LRELU)
        wp, bp = self.fwd_layer(P, "conv_last")
        self.conv(fwd, (self.V, 0), (self.Y, 0), wp, bp, B, 4 * h, 4 * w, nf, out_ch)
        self.fwd_ops = self.freeze(fwd)
        self.fwd_meta = fwd.meta
        self.bwd_ops = None
        self.param_names = [n for n in P]
        self._n_fwd_layers = len(self.pack_layers)      # the layers after these belong to backward
        if need_grad:
            self._build_backward(P, Rb, flat_grad)
        self.finalize_packing()
        self.finalize_unpack()

    def _build_backward(self, P, Rb, flat_grad):
        B, h, w, nf, gc, nb = self.B, self.h, self.w, self.nf, self.gc, self.nb
        in_ch, out_ch = self.in_ch, self.out_ch
        nR = 3 * nb
        g8 = 4 * (gc // 8)
        ctot = nf + 4 * gc
        H4, W4 = 4 * h, 4 * w
        bwd = OpList()
        self.setup_grads(P, flat_grad)
        self.memset(bwd, self.stage)
        self.GY = self.buf(B, out_ch, H4, W4)           # dL/d(out): filled by the caller / loss kernel
        self.gV = self.buf(B, nf, H4, W4)
        self.gU1 = self.buf(B, nf, H4, W4)
        self.gyu1 = self.buf(B, 4 * nf, 2 * h, 2 * w)
        self.gU0 = self.buf(B, nf, 2 * h, 2 * w)
        self.gyu0 = self.buf(B, 4 * nf, h, w)
        self.gT1 = self.buf(B, nf, h, w)
        self.G = [self.buf(B, ctot, h, w) for _ in range(nR)]
        self.gfeat = self.buf(B, nf, h, w)
        self.gX0 = self.buf(B, in_ch, h, w) if self.need_input_grad else None

        side_marks: List[Tuple[int, OpList, List[str]]] = []

        def aside(emit, names):
            """Record a weight-gradient launch that may start once every chain op recorded so
            far has run; names: the parameters whose gradients it completes."""
            if self.overlap_wgrad:
                ops = OpList()
                emit(ops)
                side_marks.append((len(bwd), ops, names))
            else:
                emit(bwd)

        wb = lambda layer: [layer + ".weight", layer + ".bias"]

        # conv_last: GY -> gV, masked by LeakyReLU'(V)               (esrgan/generator.py:71)
        aside(lambda o: self.layer_wgrad(o, P, "conv_last", (self.V, 0), (self.GY, 0), B, H4, W4), wb("conv_last"))
        self.conv(bwd, (self.GY, 0), (self.gV, 0), self.dgrad_layer(P, "conv_last", nf), 0, B, H4, W4,
                  out_ch, nf, mask=(self.V, 0), mask_slope=LRELU)
        # conv_hr: gV -> gU1, masked by LeakyReLU'(U1) (activation of upsample.1, :48)
        aside(lambda o: self.layer_wgrad(o, P, "conv_hr", (self.U1, 0), (self.gV, 0), B, H4, W4), wb("conv_hr"))
        self.conv(bwd, (self.gV, 0), (self.gU1, 0), self.dgrad_layer(P, "conv_hr", nf), 0, B, H4, W4,
                  nf, nf, mask=(self.U1, 0), mask_slope=LRELU)
        # upsample.1: inverse shuffle, then conv backward, masked by LeakyReLU'(U0)
        self.generic(bwd, L.OP_PIXEL_UNSHUFFLE, a=(self.gU1, 0), c=(self.gyu1, 0), B=B, C_=nf, H=2 * h, W=2 * w)
        aside(lambda o: self.layer_wgrad(o, P, "upsample.1.conv", (self.U0, 0), (self.gyu1, 0), B, 2 * h, 2 * w), wb("upsample.1.conv"))
        self.conv(bwd, (self.gyu1, 0), (self.gU0, 0), self.dgrad_layer(P, "upsample.1.conv", nf), 0, B,
                  2 * h, 2 * w, 4 * nf, nf, mask=(self.U0, 0), mask_slope=LRELU)
        # upsample.0 (its input T1 = feat + trunk has no activation)
        self.generic(bwd, L.OP_PIXEL_UNSHUFFLE, a=(self.gU0, 0), c=(self.gyu0, 0), B=B, C_=nf, H=h, W=w)
        aside(lambda o: self.layer_wgrad(o, P, "upsample.0.conv", (self.T1, 0), (self.gyu0, 0), B, h, w), wb("upsample.0.conv"))
        self.conv(bwd, (self.gyu0, 0), (self.gT1, 0), self.dgrad_layer(P, "upsample.0.conv", nf), 0, B, h, w,
                  4 * nf, nf)
        # conv_body: gT1 -> gradient of the last RRDB's output
        aside(lambda o: self.layer_wgrad(o, P, "conv_body", (Rb(nR), 0), (self.gT1, 0), B, h, w), wb("conv_body"))
        last_dst = (self.G[nR - 1], g8) if nR else (self.gfeat, 0)
        self.conv(bwd, (self.gT1, 0), last_dst, self.dgrad_layer(P, "conv_body", nf), 0, B, h, w, nf, nf)

        # Dense blocks, last to first.  Their weight gradients go out in launches of several
        # blocks each: long-running workgroups, few pixel slices (little atomic traffic), block
        # order keeps one block's buffers hot in L2.
        # launch sizes (dense blocks per launch), in backward order.  The last launches run after
        # the data-gradient chain has ended, with nothing to hide behind: they shrink (and take
        # more pixel slices, below) so that this tail is short.
        if self.overlap_wgrad and self.trunk_wgrad_rdbs > 0:
            tail = [s_ for s_ in self.trunk_wgrad_tail if s_ < self.trunk_wgrad_rdbs]
            sizes, left = [], nR
            while left > sum(tail) + self.trunk_wgrad_rdbs:
                sizes.append(self.trunk_wgrad_rdbs)
                left -= self.trunk_wgrad_rdbs
            for s_ in [left - sum(tail)] + tail if left > sum(tail) else [left]:
                if s_ > 0:
                    sizes.append(s_)
        else:
            sizes = [nR]
        assert sum(sizes) == nR
        launch_sizes = iter(sizes)
        per_launch = next(launch_sizes, nR)
        pending: list = []
        pending_names: List[str] = []
        n_pending = 0

        n_launches, launch_no = len(sizes), 0

        def flush():
            nonlocal pending, pending_names, n_pending, launch_no
            if not pending:
                return
            tasks, nblocks, names = pending, n_pending, pending_names
            pending, pending_names, n_pending = [], [], 0
            per_block = len(tasks) // nblocks
            # enough pixel slices to give every CU a workgroup, at least the configured number
            if self.use_fused_rdb_bwd(B, h, w, nf, gc):
                nsl = max(self.fused_bwd_wgrad_slices, min(32, self.fused_bwd_wgrad_target_wgs // max(1, len(tasks))))
            else:
                nsl = max(self.trunk_wgrad_slices, min(32, self.trunk_wgrad_target_wgs // max(1, len(tasks))))
            launch_no += 1
            k_walk = self.trunk_wgrad_walk_last
            if k_walk is None:
                k_walk = 8 if B * h * w >= 200_000 else 2
            walk = launch_no > n_launches - k_walk
            aside(lambda o: self.wgrad(o, None, None, tasks, B, h, w, nslices=nsl,
                                       x_img_stride=self.R[0].img_stride, gy_img_stride=self.G[0].img_stride,
                                       group=per_block if per_block * nblocks == len(tasks) else 0,
                                       tile_walk=walk), names)

        for r in range(nR - 1, -1, -1):
            blk, q = divmod(r, 3)
            dst = (self.G[r - 1], g8) if r > 0 else (self.gfeat, 0)
            self.emit_rdb_bwd(bwd, P, f"body.{blk}.rdb{q + 1}.", Rb(r), self.G[r], dst, B, h, w, nf, gc, q,
                              (self.G[r + 2], g8) if q == 0 else None, defer_wgrad=pending)
            pending_names += [f"body.{blk}.rdb{q + 1}.conv{k}.{t}" for k in range(1, 6) for t in ("weight", "bias")]
            n_pending += 1
            if n_pending == per_launch:
                flush()
                per_launch = next(launch_sizes, nR)
        flush()
        # feat also feeds the trunk skip (esrgan/generator.py:68): gfeat += gT1
        self.generic(bwd, L.OP_AXPBY, a=(self.gT1, 0), c=(self.gfeat, 0), f0=1.0, f1=1.0,
                     n=self.gfeat.t.numel())
        aside(lambda o: self.layer_wgrad(o, P, "conv_first", (self.X0, 0), (self.gfeat, 0), B, h, w), wb("conv_first"))
        if self.need_input_grad:
            self.conv(bwd, (self.gfeat, 0), (self.gX0, 0), self.dgrad_layer(P, "conv_first", in_ch), 0,
                      B, h, w, nf, in_ch)
        self.trim_stage_memset(bwd)
        if self.overlap_wgrad:
            # every stream a step uses exists before the step's first launch, created in one fixed
            # order (the caller's streams, then this one): never in the middle of a backward pass
            side_stream(self.device)
            self.bwd_segments, self.bwd_ops, self.bwd_meta, seg_names, self.bwd_seg_meta = \
                self.split_ops(bwd, side_marks)
            self.bwd_seg_ranges = self._ranges_for(seg_names)
        else:
            self.bwd_ops = self.freeze(bwd)
            self.bwd_meta = bwd.meta

    # ------------------------------------------------------------------ API
    def forward_c8(self, x_nchw: torch.Tensor) -> Buf:
        """Pack weights, convert the input and run the forward ops; returns Y (C8)."""
        nf_ = self._n_fwd_layers
        if self.bwd_segments is not None and len(self.pack_layers) > nf_:
            # the data-gradient weights (transposed, tap-flipped stacks) are first needed when
            # backward starts: packed on the side stream, under the forward pass
            main, side = torch.cuda.current_stream(self.device), side_stream(self.device)
            if self._pack_evs is None:
                self._pack_evs = (torch.cuda.Event(), torch.cuda.Event())
            ready, done = self._pack_evs
            ready.record(main)                       # parameters are final here (optimizer step is behind us)
            self.run_packing(0, nf_)
            side.wait_event(ready)
            self.run_packing(nf_, None, stream=side)
            done.record(side)
            self._bwd_pack_pending = True
        else:
            self.run_packing()
        self.nchw_to_c8(x_nchw, (self.X0, 0))
        self.run(self.fwd_ops)
        return self.Y

    def forward(self, x_nchw: torch.Tensor) -> torch.Tensor:
        return self.c8_to_nchw((self.forward_c8(x_nchw), 0), self.out_ch)

    def backward(self, gout_nchw: torch.Tensor) -> Optional[torch.Tensor]:
        self.nchw_to_c8(gout_nchw, (self.GY, 0))
        self.backward_c8()
        if self.need_input_grad:
            return self.c8_to_nchw((self.gX0, 0), self.in_ch)
        return None


class BlockPlan(_GenBase):
    """Stand-alone ResidualDenseBlock / RRDB / UpsampleBlock (esrgan/generator.py:6-48)."""

    def __init__(self, kind: str, params: Dict[str, torch.Tensor], B: int, h: int, w: int, tdtype, *,
                 nf=64, gc=32, need_grad=True, need_input_grad=True):
        dev = next(iter(params.values())).device
        super().__init__(dev, tdtype)
        if nf % 8 or gc % 8:
            raise ValueError("channel counts must be multiples of 8")
        self.kind, self.B, self.h, self.w, self.nf, self.gc = kind, B, h, w, nf, gc
        self.need_grad, self.need_input_grad = need_grad, need_input_grad
        P = params
        self.param_names = [n for n in P]
        ctot, g8 = nf + 4 * gc, 4 * (gc // 8)
        fwd: list = []
        bwd: list = []
        if need_grad:
            self.setup_grads(P)
            self.memset(bwd, self.stage)
        if kind in ("rdb", "rrdb"):
            pres = [""] if kind == "rdb" else ["rdb1.", "rdb2.", "rdb3."]
            n = len(pres)
            self.R = [self.buf(B, ctot, h, w) for _ in range(n)]
            self.OUT = self.buf(B, nf, h, w)
            self.in_view: View = (self.R[0], 0)
            self.out_ch = nf
            for i, pre in enumerate(pres):
                q = i if kind == "rrdb" else 0
                dst = (self.R[i + 1], 0) if i + 1 < n else (self.OUT, 0)
                self.emit_rdb_fwd(fwd, P, pre, self.R[i], dst, B, h, w, nf, gc, q,
                                  (self.R[0], 0) if q == 2 else None)
            if need_grad:
                self.G = [self.buf(B, ctot, h, w) for _ in range(n)]
                self.gX = self.buf(B, nf, h, w)
                self.gout_view: View = (self.G[n - 1], g8)
                for i in range(n - 1, -1, -1):
                    q = i if kind == "rrdb" else 1      # a lone block: plain "+ g_out" skip
                    dst = (self.G[i - 1], g8) if i > 0 else (self.gX, 0)
                    self.emit_rdb_bwd(bwd, P, pres[i], self.R[i], self.G[i], dst, B, h, w, nf, gc, q,
                                      (self.G[2], g8) if (kind == "rrdb" and q == 0) else None)
        elif kind == "upsample":
            self.X = self.buf(B, nf, h, w)
            self.OUT = self.buf(B, nf, 2 * h, 2 * w)
            self.in_view = (self.X, 0)
            self.out_ch = nf
            wp, bp = self.fwd_layer(P, "conv", perm=1)
            self.conv(fwd, (self.X, 0), (self.OUT, 0), wp, bp, B, h, w, nf, 4 * nf, act=LRELU, ps=1)
            if need_grad:
                self.gOUT = self.buf(B, nf, 2 * h, 2 * w)
                self.gy = self.buf(B, 4 * nf, h, w)
                self.gX = self.buf(B, nf, h, w)
                self.gout_view = (self.gOUT, 0)
                # LeakyReLU'(out) on the incoming gradient, inverse shuffle, conv backward
                self.generic(bwd, L.OP_ACT_BWD, a=(self.gOUT, 0), mask=(self.OUT, 0), c=(self.gOUT, 0),
                             f0=LRELU, n=self.gOUT.t.numel())
                self.generic(bwd, L.OP_PIXEL_UNSHUFFLE, a=(self.gOUT, 0), c=(self.gy, 0), B=B, C_=nf, H=h, W=w)
                self.layer_wgrad(bwd, P, "conv", (self.X, 0), (self.gy, 0), B, h, w)
                self.conv(bwd, (self.gy, 0), (self.gX, 0), self.dgrad_layer(P, "conv", nf), 0, B, h, w, 4 * nf, nf)
        else:
            raise ValueError(kind)
        self.fwd_ops = self.freeze(fwd)
        self.trim_stage_memset(bwd)
        self.bwd_ops = self.freeze(bwd) if need_grad else None
        self.finalize_packing()
        self.finalize_unpack()

    def forward(self, x_nchw: torch.Tensor) -> torch.Tensor:
        self.run_packing()
        self.nchw_to_c8(x_nchw, self.in_view)
        self.run(self.fwd_ops)
        return self.c8_to_nchw((self.OUT, 0), self.out_ch)

    def backward(self, gout_nchw: torch.Tensor) -> Optional[torch.Tensor]:
        self.nchw_to_c8(gout_nchw, self.gout_view)
        self.backward_c8()
        return self.c8_to_nchw((self.gX, 0), self.nf)
