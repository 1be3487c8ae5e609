"""ctypes binding of libsrx.so (the C ABI declared in include/srx.h).

The product path has no fallback: if the shared library is missing or a call
fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os

# PyTorch's ROCm wheel bundles its own HIP runtime (torch/lib/libamdhip64.so).  It must be the
# one already mapped when libsrx.so is opened, so that our kernels and torch's allocator/streams
# share a single runtime instance; importing torch first guarantees that order.
import torch  # noqa: F401  (import order matters)

_HERE = os.path.dirname(os.path.abspath(__file__))
# SRX_LIB_PATH: another build of the library (A/B measurements of kernel variants)
LIB_PATH = os.environ.get("SRX_LIB_PATH") or os.path.join(_HERE, "libsrx.so")

BF16, F32 = 0, 1

OP_CONV, OP_WGRAD, OP_PIXEL_UNSHUFFLE, OP_CHANNEL_SUM, OP_ZERO_STUFF = 1, 2, 3, 4, 5
OP_MAXPOOL_FWD, OP_MAXPOOL_BWD, OP_MEMSET, OP_AXPBY, OP_GAP_FWD, OP_GAP_BWD = 6, 7, 8, 9, 10, 11
OP_ACT_BWD = 12
OP_RDB_FWD = 13
OP_RDB_BWD = 14

vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class PackSeg(C.Structure):
    _fields_ = [("src", vp), ("cin_src", i32), ("cout_src", i32), ("c_begin", i32), ("c_count", i32),
                ("src_off", i32), ("src_o_off", i32), ("transposed", i32), ("scale", f32)]


class PackLayer(C.Structure):
    _fields_ = [("dst_w", vp), ("dst_bias", vp), ("src_bias", vp), ("cout", i32), ("cin", i32),
                ("nblk", i32), ("nchunk", i32), ("taps", i32), ("perm", i32), ("seg_begin", i32),
                ("seg_count", i32)]


class ConvDesc(C.Structure):
    _fields_ = [("dtype", i32), ("ksize", i32), ("stride", i32), ("B", i32), ("H", i32), ("W", i32),
                ("cin", i32), ("cout", i32), ("pixel_shuffle", i32),
                ("x", vp), ("x_img_stride", i64), ("y", vp), ("y_img_stride", i64),
                ("w", vp), ("bias", vp),
                ("mask", vp), ("mask_img_stride", i64), ("mask_slope", f32),
                ("act_slope", f32), ("alpha", f32),
                ("res1", vp), ("res1_img_stride", i64), ("beta1", f32),
                ("res2", vp), ("res2_img_stride", i64), ("beta2", f32)]


class WgradTask(C.Structure):
    _fields_ = [("x", vp), ("gy", vp), ("x_oct", i32), ("gy_oct", i32), ("x_noct", i32), ("gy_noct", i32),
                ("x_skip", i32), ("gy_skip", i32), ("out", vp * 4), ("ld", i32 * 4), ("co_n", i32 * 4), ("ci0", i32 * 4),
                ("co0", i32 * 4), ("scale", f32 * 4),
                ("bias_out", vp * 2), ("bias_co0", i32 * 2), ("bias_n", i32 * 2), ("bias_scale", f32 * 2)]


class WgradDesc(C.Structure):
    _fields_ = [("dtype", i32), ("ksize", i32), ("stride", i32), ("B", i32), ("H", i32), ("W", i32),
                ("x", vp), ("x_img_stride", i64), ("gy", vp), ("gy_img_stride", i64),
                ("tasks", vp), ("ntasks", i32), ("nslices", i32), ("group", i32), ("tile_walk", i32), ("pad_", i32),
                ("slice_stride", i64)]


class UnpackItem(C.Structure):
    _fields_ = [("src", vp), ("dst", vp), ("cout", i32), ("cin", i32), ("taps", i32),
                ("accumulate", i32), ("nslices", i32), ("slice_stride", i64)]


class SnLayer(C.Structure):
    _fields_ = [("w", vp), ("u", vp), ("v", vp), ("inv_sigma", vp), ("wv", vp),
                ("rows", i32), ("cols", i32)]


class SnBwdLayer(C.Structure):
    _fields_ = [("w", vp), ("u", vp), ("v", vp), ("inv_sigma", vp), ("dw", vp), ("dot", vp),
                ("rows", i32), ("cols", i32)]


class GenericArgs(C.Structure):
    _fields_ = [("dtype", i32), ("B", i32), ("C", i32), ("H", i32), ("W", i32), ("noct", i32),
                ("a", vp), ("a_img_stride", i64), ("b", vp), ("b_img_stride", i64),
                ("c", vp), ("c_img_stride", i64), ("mask", vp), ("mask_img_stride", i64),
                ("f0", f32), ("f1", f32), ("n", i64)]


class Tile(C.Structure):
    _fields_ = [("y0", i32), ("x0", i32), ("th", i32), ("tw", i32), ("offset", i64)]


class Crop(C.Structure):
    _fields_ = [("img", vp), ("H", i32), ("W", i32), ("y0", i32), ("x0", i32), ("flags", i32), ("pad_", i32)]


class RdbDesc(C.Structure):
    _fields_ = [("dtype", i32), ("B", i32), ("H", i32), ("W", i32), ("nf", i32), ("gc", i32),
                ("store_growth", i32), ("tile", i32),
                ("buf", vp), ("buf_img_stride", i64), ("y", vp), ("y_img_stride", i64),
                ("w", vp), ("bias", vp), ("act_slope", f32), ("alpha", f32), ("beta1", f32), ("beta2", f32),
                ("res2", vp), ("res2_img_stride", i64), ("act", vp), ("act_img_stride", i64)]


class _OpU(C.Union):
    _fields_ = [("conv", ConvDesc), ("wgrad", WgradDesc), ("g", GenericArgs), ("rdb", RdbDesc)]


class Op(C.Structure):
    _fields_ = [("kind", i32), ("pad_", i32), ("u", _OpU)]


_SIGS = {
    "srx_version": (C.c_int, []),
    "srx_last_error": (C.c_char_p, []),
    "srx_sizeof": (C.c_int, [C.c_int]),
    "srx_selftest_per_device_once": (C.c_int, []),
    "srx_nchw_to_c8": (C.c_int, [C.c_int, vp, vp, i64, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "srx_c8_to_nchw": (C.c_int, [C.c_int, vp, i64, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "srx_pack_weights": (C.c_int, [C.c_int, vp, vp, C.c_int, i64, vp]),
    "srx_pack_weights_scaled": (C.c_int, [C.c_int, vp, vp, vp, C.c_int, i64, vp]),
    "srx_pack_weights_flat": (C.c_int, [C.c_int, vp, vp, vp, vp, C.c_int, vp]),
    "srx_conv2d": (C.c_int, [C.POINTER(ConvDesc), vp]),
    "srx_conv2d_wgrad": (C.c_int, [C.POINTER(WgradDesc), vp]),
    "srx_rdb_forward": (C.c_int, [C.POINTER(RdbDesc), vp]),
    "srx_rdb_backward": (C.c_int, [C.POINTER(RdbDesc), vp]),
    "srx_unpack_wgrad": (C.c_int, [vp, C.c_int, i64, vp]),
    "srx_channel_sum": (C.c_int, [C.c_int, vp, i64, C.c_int, C.c_int, C.c_int, C.c_int, vp, f32, vp]),
    "srx_pixel_unshuffle2": (C.c_int, [C.c_int, vp, i64, vp, i64, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "srx_zero_stuff2": (C.c_int, [C.c_int, vp, i64, vp, i64, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "srx_maxpool2_fwd": (C.c_int, [C.c_int, vp, i64, vp, i64, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "srx_maxpool2_bwd": (C.c_int, [C.c_int, vp, i64, vp, i64, vp, i64, vp, i64, f32,
                                   C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "srx_gap_fwd": (C.c_int, [C.c_int, vp, i64, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "srx_gap_bwd": (C.c_int, [C.c_int, vp, vp, i64, vp, i64, f32, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "srx_axpby": (C.c_int, [C.c_int, vp, vp, f32, f32, i64, vp]),
    "srx_act_bwd": (C.c_int, [C.c_int, vp, vp, vp, f32, i64, vp]),
    "srx_l1_c8_nchw": (C.c_int, [C.c_int, vp, i64, vp, vp, vp, i64, f32,
                                 C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "srx_l1_c8": (C.c_int, [C.c_int, vp, i64, vp, i64, vp, vp, i64, f32, f32,
                            C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "srx_bce_logits": (C.c_int, [C.c_int, vp, C.c_int, f32, f32, vp, vp, f32, vp]),
    "srx_sqerr_c8_nchw": (C.c_int, [C.c_int, vp, i64, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "srx_sn_power_iter": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "srx_sn_power_iter_noalloc": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "srx_sn_scratch_floats": (C.c_int, []),
    "srx_sn_backward": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp]),
    "srx_adam": (C.c_int, [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, C.c_int, f32, vp]),
    "srx_run_ops": (C.c_int, [vp, C.c_int, vp]),
    "srx_stream_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
    "srx_stream_destroy": (C.c_int, [vp]),
    "srx_tile_gather": (C.c_int, [C.c_int, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp, i64, vp]),
    "srx_crop_augment": (C.c_int, [vp, C.c_int, C.c_int, vp, vp]),
    "srx_tile_blend": (C.c_int, [C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp,
                                 C.c_int, C.c_int, C.c_int, C.c_int, vp]),
}

EXPORTS = tuple(_SIGS)

_lib = None


def lib():
    """Load libsrx.so (once).  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
                "(or python -c 'import __graft_entry__ as g; g.build()'). There is no fallback path.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        structs = (ConvDesc, WgradTask, WgradDesc, PackSeg, PackLayer, UnpackItem, SnLayer,
                   SnBwdLayer, GenericArgs, Op, Tile, Crop, RdbDesc)
        for i, st in enumerate(structs):
            if l.srx_sizeof(i) != C.sizeof(st):
                raise RuntimeError(f"libsrx ABI mismatch: sizeof({st.__name__}) is {C.sizeof(st)} in "
                                   f"the binding but {l.srx_sizeof(i)} in {LIB_PATH}")
        _lib = l
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().srx_last_error().decode(errors="replace")
        raise RuntimeError(f"libsrx {what} failed (status {rc}): {msg}")
