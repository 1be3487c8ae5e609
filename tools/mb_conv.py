import importlib, sys, torch, time
sys.path.insert(0, '.')
E = importlib.import_module("super-resolution_amd.engine")
L = importlib.import_module("super-resolution_amd._lib")
dev = torch.device("cuda:0")
def bench(B, H, W, cin, cout, dt=torch.bfloat16, reps=200, stride=1, distinct=True):
    plan = E.Plan(dev, dt)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    bias = torch.zeros(cout, device=dev)
    nbuf = 8 if distinct else 1
    X = [plan.buf(B, cin, H, W, zero=True) for _ in range(nbuf)]
    for x in X: x.t.normal_()
    Y = plan.buf(B, cout, H // stride, W // stride)
    wp, bp = plan.packed_layer(cout, cin, 9, 0, [E.conv_seg(w)], bias.data_ptr())
    plan.finalize_packing(); plan.run_packing()
    ops = []
    for i in range(reps):
        plan.conv(ops, (X[i % nbuf], 0), (Y, 0), wp, bp, B, H, W, cin, cout, act=0.2, stride=stride)
    fr = plan.freeze(ops)
    plan.run(fr); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); plan.run(fr); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    fl = 2.0 * B * (H // stride) * (W // stride) * 9 * cin * cout
    print(f"B{B} {H}x{W} cin{cin} cout{cout} {str(dt)[6:]}: {us:8.2f} us  {fl / us / 1e6:8.1f} TF/s")
if len(sys.argv) > 1 and sys.argv[1] == "big":
    bench(64, 128, 128, 192, 64, reps=50)
    bench(64, 128, 128, 64, 64, reps=50)
    bench(64, 128, 128, 64, 256, reps=50)
    bench(16, 256, 256, 64, 256, reps=20)
    bench(16, 512, 512, 64, 64, reps=20)
    sys.exit(0)
for cin in (16, 32, 64, 96, 128, 160, 192):
    bench(16, 64, 64, cin, 32)
bench(16, 64, 64, 192, 64)
bench(16, 64, 64, 64, 256)
bench(64, 128, 128, 64, 32, reps=50)
bench(64, 128, 128, 160, 32, reps=50)
bench(64, 128, 128, 192, 64, reps=50)
bench(16, 256, 256, 64, 64, reps=50)
bench(16, 64, 64, 64, 32, torch.float32)
for cin in (64, 128, 192, 256):
    bench(16, 64, 64, cin, 64)
