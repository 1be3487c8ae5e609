"""Shared test utilities: package loading, golden fixtures, single-op runners."""
import importlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import esrgan_oracle as O  # noqa: E402

PKG = "super-resolution_amd"


def pkg(sub=""):
    return importlib.import_module(PKG + (("." + sub) if sub else ""))


_golden = None


def golden():
    global _golden
    if _golden is None:
        d = os.path.join(ROOT, "tests", "golden")
        with open(os.path.join(d, "golden.json")) as f:
            man = json.load(f)
        arr = np.load(os.path.join(d, "golden.npz"))
        _golden = (man, arr)
    return _golden


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """max |a-b| / max |b|  (b = expected)."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


TOL = {torch.float32: 1e-3, torch.bfloat16: 2e-2}


def record_observed(test: str, **values) -> None:
    """Append the error figures a test observed to gpurun_out/observed_errors.jsonl (when that
    directory exists: the GPU box's scratch output), so that tolerances can be stated next to what
    was measured.  Never fails a test."""
    try:
        d = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(d):
            with open(os.path.join(d, "observed_errors.jsonl"), "a") as f:
                f.write(json.dumps({"test": test, **{k: float(v) for k, v in values.items()}}) + "\n")
    except (OSError, TypeError, ValueError):
        pass


def full_grads_bad(named_grads, index, arr, tol):
    """Element-wise comparison of whole gradient tensors against the reference's (the
    `full_grads` index of a golden case): returns [(name, rel L2, rel max)] of the tensors whose
    relative L2 error exceeds tol or whose max-norm error exceeds 4*tol."""
    grads = dict(named_grads)
    bad = []
    assert index, "the golden case holds no full gradient tensors"
    for n, ent in index.items():
        g = grads[n].detach().float().cpu().flatten()[::ent["step"]]
        ref = torch.from_numpy(arr[ent["array"]]).flatten()
        assert g.shape == ref.shape, (n, g.shape, ref.shape)
        e2, em = rel_l2(g, ref), rel_err(g, ref)
        if not (e2 < tol and em < 4 * tol):
            bad.append((n, e2, em))
    return bad


def fill_module_(m, salt=0, gain=1.0):
    sd = m.state_dict()
    O.fill_state_dict_(sd, salt, gain)
    return m


def run_single_conv(x, w, b, *, stride=1, tdtype=torch.float32, act=1.0, alpha=1.0, res=None, beta=1.0,
                    mask=None, mask_slope=0.0, ps=0):
    """One srx_conv2d call through engine.Plan on cuda; NCHW fp32 in/out."""
    E = pkg("engine")
    L = pkg("_lib")
    dev = torch.device("cuda")
    x, w = x.to(dev), w.to(dev).contiguous()
    b = b.to(dev) if b is not None else None
    B, cin, H, W = x.shape
    cout, _, k, _ = w.shape
    plan = E.Plan(dev, tdtype)
    OH = (H + 2 * (k // 2) - k) // stride + 1
    OW = (W + 2 * (k // 2) - k) // stride + 1
    X = plan.buf(B, cin, H, W)
    if ps:
        Y = plan.buf(B, cout // 4, 2 * OH, 2 * OW)
    else:
        Y = plan.buf(B, cout, OH, OW)
    wp, bp = plan.packed_layer(cout, cin, k * k, 1 if ps else 0, [E.conv_seg(w)],
                               b.data_ptr() if b is not None else 0)
    plan.finalize_packing()
    plan.run_packing()
    plan.nchw_to_c8(x, (X, 0))
    kw = {}
    if res is not None:
        Rb = plan.buf(B, cout, OH, OW)
        plan.nchw_to_c8(res.to(dev), (Rb, 0))
        kw.update(res1=(Rb, 0), beta1=beta)
    if mask is not None:
        Mb = plan.buf(B, cout, OH, OW)
        plan.nchw_to_c8(mask.to(dev), (Mb, 0))
        kw.update(mask=(Mb, 0), mask_slope=mask_slope)
    ops = []
    plan.conv(ops, (X, 0), (Y, 0), wp, bp, B, H, W, cin, cout, ksize=k, stride=stride, act=act,
              alpha=alpha, ps=ps, **kw)
    plan.run(plan.freeze(ops))
    out = plan.c8_to_nchw((Y, 0), cout // 4 if ps else cout)
    torch.cuda.synchronize()
    return out.cpu()


def start_stream_watchdog(after_s: float, streams=None, label: str = "", exit_code=None):
    """A daemon thread that, `after_s` seconds from now, writes to stderr what the HIP runtime reports
    for every stream of interest — hipStreamQuery per stream (0 = everything queued on it has
    completed, 600 = not ready) for the null stream, torch's current stream, the library's side
    stream and any stream in `streams` — plus faulthandler's traceback of every Python thread.
    A process whose main thread sits in a blocking call (synchronize, a synchronous copy) can thus
    still say whether the DEVICE never finished the work (a stream reports 600) or the host wait
    never returned although it did (every stream reports 0).  The query calls do not block.
    exit_code: if given, the process ends with it after the dump.  Returns a function that
    cancels the watchdog."""
    import ctypes as C
    import faulthandler
    import threading
    cancelled = threading.Event()

    def run():
        if cancelled.wait(after_s):
            return
        rep = {"watchdog": label, "after_s": after_s}
        try:
            hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
            q = lambda ptr: int(hip.hipStreamQuery(C.c_void_p(ptr)))
            rep["null_stream"] = q(0)
            named = dict(streams or {})
            E = sys.modules.get(PKG + ".engine")
            if E is not None:
                for idx, st in getattr(E, "_SIDE_STREAMS", {}).items():
                    named["side_stream_dev%d" % idx] = st
            for k, st in named.items():
                rep[k] = q(st.cuda_stream)
        except Exception as e:                     # the dump must not raise out of the thread
            rep["error"] = repr(e)
        sys.stderr.write("STREAM-WATCHDOG " + json.dumps(rep) + "\n")
        sys.stderr.flush()
        faulthandler.dump_traceback(all_threads=True)
        if exit_code is not None:
            os._exit(exit_code)

    threading.Thread(target=run, daemon=True).start()
    return cancelled.set
