"""mb_shapes.py B,H,W,cin,cout[,reps] ... : stand-alone time of srx_conv2d (3x3, stride 1, bf16) on each shape
(eight distinct inputs in rotation, so the activations are not warm in L2 from the previous launch)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
E = importlib.import_module("super-resolution_amd.engine")
dev = torch.device("cuda:0")


def bench(B, H, W, cin, cout, reps=100, dt=torch.bfloat16):
    plan = E.Plan(dev, dt)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    bias = torch.zeros(cout, device=dev)
    X = [plan.buf(B, cin, H, W, zero=True) for _ in range(8)]
    for x in X:
        x.t.normal_()
    Y = plan.buf(B, cout, H, W)
    # MB_WEIGHT_SETS=n: rotate among n packed copies of the weights (every launch of a real layer
    # chain reads weights no earlier launch has touched)
    nw = int(os.environ.get("MB_WEIGHT_SETS", "1"))
    packed = [plan.packed_layer(cout, cin, 9, 0, [E.conv_seg(w)], bias.data_ptr()) for _ in range(nw)]
    plan.finalize_packing(); plan.run_packing()
    ops = []
    for i in range(reps):
        wp, bp = packed[i % nw]
        plan.conv(ops, (X[i % 8], 0), (Y, 0), wp, bp, B, H, W, cin, cout, act=0.2)
    fr = plan.freeze(ops)
    plan.run(fr); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); plan.run(fr); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    fl = 2.0 * B * H * W * 9 * cin * cout
    print(f"B{B} {H}x{W} {cin}->{cout}: {best:8.2f} us  {fl / best / 1e6:8.1f} TF/s  sum {float(Y.t.float().sum()):.4e}", flush=True)


for a in sys.argv[1:]:
    v = [int(t) for t in a.split(",")]
    bench(*v)
