"""One convolution shape, repeated: the target of counter passes."""
import importlib, sys, torch
sys.path.insert(0, '.')
E = importlib.import_module("super-resolution_amd.engine")
dev = torch.device("cuda:0")
cin = int(sys.argv[1]) if len(sys.argv) > 1 else 192
cout = int(sys.argv[2]) if len(sys.argv) > 2 else 32
B, H, W, reps = 16, 64, 64, 50
plan = E.Plan(dev, torch.bfloat16)
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
