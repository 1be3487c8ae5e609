"""Stand-alone time of the HBM-bound helper kernels at the sizes of the benchmark's training step."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
E = importlib.import_module("super-resolution_amd.engine")
L = importlib.import_module("super-resolution_amd._lib")
dev = torch.device("cuda:0")
B, h, w, nf = 16, 64, 64, 64
plan = E.Plan(dev, torch.bfloat16)


def timeit(name, ops, nbytes, reps=20):
    fr = plan.freeze(ops * reps)
    plan.run(fr); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); plan.run(fr); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    print(f"{name:28s} {best:8.2f} us  {nbytes / best / 1e6:6.2f} TB/s", flush=True)


for (hh, ww) in ((2 * h, 2 * w), (h, w)):
    gU = plan.buf(B, nf, 2 * hh, 2 * ww); gU.t.normal_()
    gy = plan.buf(B, 4 * nf, hh, ww)
    ops = []
    plan.generic(ops, L.OP_PIXEL_UNSHUFFLE, a=(gU, 0), c=(gy, 0), B=B, C_=nf, H=hh, W=ww)
    timeit(f"pixel_unshuffle2 -> {hh}x{ww}", ops, 2 * gU.nbytes)
a = plan.buf(B, nf, h, w); a.t.normal_(); c = plan.buf(B, nf, h, w); c.t.normal_()
ops = []
plan.generic(ops, L.OP_AXPBY, a=(a, 0), c=(c, 0), f0=1.0, f1=1.0, n=c.t.numel())
timeit("axpby 64ch 64x64", ops, 3 * a.nbytes)
