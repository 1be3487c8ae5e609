import ctypes, importlib, sys, torch
import numpy as np
sys.path.insert(0, '.')
L = importlib.import_module("super-resolution_amd._lib")
L.LIB_PATH = "scratch/libsrx_STAMP.so"
E = importlib.import_module("super-resolution_amd.engine")
dev = torch.device("cuda:0")
lib = L.lib()
lib.srx_dbg_stamps.argtypes = [ctypes.c_void_p]
def run(nconv, B=16, H=64, W=64):
    plan = E.Plan(dev, torch.bfloat16)
    R = plan.buf(B, 192, H, W, zero=True); R.t.normal_()
    layers = []
    for k in range(1, nconv + 1):
        cin = 64 + 32 * (k - 1)
        w = torch.randn(32, cin, 3, 3, device=dev) * 0.03
        layers.append((cin, plan.packed_layer(32, cin, 9, 0, [E.conv_seg(w)], torch.zeros(32, device=dev).data_ptr())))
    plan.finalize_packing(); plan.run_packing()
    ops = []
    for rep in range(6):
        for k, (cin, (wp, bp)) in enumerate(layers):
            plan.conv(ops, (R, 0), (R, 8 + 4 * k), wp, bp, B, H, W, cin, 32, act=0.2)
    fr = plan.freeze(ops)
    plan.run(fr); torch.cuda.synchronize()
    plan.run(fr); torch.cuda.synchronize()
    buf = np.zeros(1024 * 8, dtype=np.uint64)
    assert lib.srx_dbg_stamps(buf.ctypes.data) == 0
    cyc = np.zeros(1024 * 8, dtype=np.uint64)
    lib.srx_dbg_cycles.argtypes = [ctypes.c_void_p]
    assert lib.srx_dbg_cycles(cyc.ctypes.data) == 0
    nwg = B * (H // 8) * (W // 32)
    cy = cyc.reshape(1024, 8)[:min(nwg, 1024), :].astype(np.int64)
    rt = buf.reshape(1024, 8)[:min(nwg, 1024), :].astype(np.int64)
    dcy, drt = cy[:, 5] - cy[:, 4], (rt[:, 5] - rt[:, 4]) * 10e-9
    print("cycles waiting at [vmcnt + barrier], summed over chunks: median %d of loop %d" % (int(np.median(cy[:, 1])), int(np.median(dcy))))
    print("in-loop clock (chunk0 landed -> loop done): median %.2f GHz" % float(np.median(dcy / drt) / 1e9),
          " loop cycles median", int(np.median(dcy)), " per chunk", int(np.median(dcy)) // max(1, layers[-1][0] // 16 - 0))
    st = buf.reshape(1024, 8)[:min(nwg, 1024), :].astype(np.int64)

    t0 = st[:, 0].min()
    st = (st - t0) * 10  # ns
    names = ["entry", "decoded+bias ld", "offsets done", "issued+setup", "chunk0 landed", "loop done", "pass1 done", "stores done"]
    print(f"--- last conv cin={layers[-1][0]} ({layers[-1][0]//16} chunks), {nwg} WGs; ns since first WG entry")
    for i, n in enumerate(names):
        c = st[:, i]
        print(f"{n:16s} min {c.min():6d}  median {int(np.median(c)):6d}  p90 {int(np.percentile(c,90)):6d}  max {c.max():6d}")
    d = st[:, 1:] - st[:, :-1]
    print("per-WG phase medians (ns):", [int(np.median(d[:, i])) for i in range(7)])
for n in (1, 4):
    run(n)
run(4, 64, 128, 128)
