import importlib, sys, torch, os
sys.path.insert(0, '.')
E = importlib.import_module("super-resolution_amd.engine")
dev = torch.device("cuda:0")
def bench(B, H, W, cin, cout, reps=200):
    plan = E.Plan(dev, torch.bfloat16)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    X = [plan.buf(B, cin, H, W, zero=True) for _ in range(8)]
    for x in X: x.t.normal_()
    Y = plan.buf(B, cout, H, W)
    wp, bp = plan.packed_layer(cout, cin, 9, 0, [E.conv_seg(w)], 0)
    plan.finalize_packing(); plan.run_packing()
    ops = []
    for i in range(reps):
        plan.conv(ops, (X[i % 8], 0), (Y, 0), wp, bp, B, H, W, cin, cout, act=0.2)
    fr = plan.freeze(ops)
    plan.run(fr); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); plan.run(fr); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
print("debug", os.environ.get("SRX_CONV_DEBUG"), " ".join(f"cin{c}:{bench(16,64,64,c,32):.2f}us" for c in (16, 64, 160)), f"c192x64:{bench(16,64,64,192,64):.2f}us")
