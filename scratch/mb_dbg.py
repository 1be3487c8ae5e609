import importlib, sys, torch
sys.path.insert(0, '.')
L = importlib.import_module("super-resolution_amd._lib")
if len(sys.argv) > 1 and sys.argv[1] != "base":
    L.LIB_PATH = "scratch/libsrx_%s.so" % sys.argv[1]
E = importlib.import_module("super-resolution_amd.engine")
dev = torch.device("cuda:0")
def bench(B, H, W, cin, cout, dt=torch.bfloat16, reps=200):
    plan = E.Plan(dev, dt)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    bias = torch.zeros(cout, device=dev)
    X = [plan.buf(B, cin, H, W, zero=True) for _ in range(8)]
    for x in X: x.t.normal_()
    Y = plan.buf(B, cout, H, W)
    wp, bp = plan.packed_layer(cout, cin, 9, 0, [E.conv_seg(w)], bias.data_ptr())
    plan.finalize_packing(); plan.run_packing()
    ops = []
    for i in range(reps):
        plan.conv(ops, (X[i % 8], 0), (Y, 0), wp, bp, B, H, W, cin, cout, act=0.2)
    fr = plan.freeze(ops)
    plan.run(fr); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); plan.run(fr); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
r = [bench(16, 64, 64, cin, 32) for cin in (16, 64, 128, 192)] + [bench(16, 64, 64, 192, 64), bench(64, 128, 128, 192, 64, reps=30)]
print(sys.argv[1:], " ".join(f"{u:8.2f}" for u in r))
