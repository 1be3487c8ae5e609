"""roctx ranges around the phases of a step (forward, loss, backward, exchange, optimizer), so a
``rocprofv3 --marker-trace`` timeline shows them next to the kernels.  The reference has no
profiling hooks at all (no torch.profiler, no NVTX: only tqdm progress bars,
esrgan/train_esrgan.py:82); these are an addition and cost nothing unless switched on.

Off by default.  ``SRX_ROCTX=1`` in the environment (read once) or ``enable(True)`` turns them
on; the markers then go to ``libroctx64.so`` of the ROCm installation.  If that library cannot be
loaded, enabling raises: a silently missing trace is worse than an error.
"""
from __future__ import annotations

import contextlib
import ctypes
import os

_lib = None
_enabled = False


def _load():
    global _lib
    if _lib is None:
        for name in ("libroctx64.so", "/opt/rocm/lib/libroctx64.so", "librocprofiler-sdk-roctx.so"):
            try:
                _lib = ctypes.CDLL(name)
                break
            except OSError:
                continue
        if _lib is None:
            raise RuntimeError("SRX_ROCTX is set but libroctx64.so could not be loaded")
        _lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
        _lib.roctxRangePushA.restype = ctypes.c_int
        _lib.roctxRangePop.restype = ctypes.c_int
    return _lib


def enable(on: bool = True) -> None:
    global _enabled
    if on:
        _load()
    _enabled = bool(on)


def enabled() -> bool:
    return _enabled


def push(name: str) -> None:
    if _enabled:
        _lib.roctxRangePushA(name.encode())


def pop() -> None:
    if _enabled:
        _lib.roctxRangePop()


@contextlib.contextmanager
def range_(name: str):
    """``with trace.range_("backward"): ...`` — a no-op unless tracing is enabled."""
    if not _enabled:
        yield
        return
    _lib.roctxRangePushA(name.encode())
    try:
        yield
    finally:
        _lib.roctxRangePop()


if os.environ.get("SRX_ROCTX", "") not in ("", "0"):
    enable(True)
