"""mb_block.py B H W [group sizes...]: forward of NBLK chained dense blocks (5 convolutions each, the
trunk's shapes) with the batch taken whole, or in groups of images that run the whole chain one after the
other (activations of a group stay in the memory-side cache between the convolutions that re-read them)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
E = importlib.import_module("super-resolution_amd.engine")
dev = torch.device("cuda:0")
B, H, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
groups = [int(a) for a in sys.argv[4:]] or [B]
NBLK, dt = 6, torch.bfloat16
plan = E.Plan(dev, dt)
R = [plan.buf(B, 192, H, W, zero=True) for _ in range(NBLK + 1)]
R[0].t.normal_()
layers = []
for r in range(NBLK):
    for k in range(5):
        cin, cout = 64 + 32 * k, (32 if k < 4 else 64)
        w = torch.randn(cout, cin, 3, 3, device=dev) * (0.3 / (cin * 9) ** 0.5)
        layers.append(plan.packed_layer(cout, cin, 9, 0, [E.conv_seg(w)], torch.zeros(cout, device=dev).data_ptr()) + (cin, cout))
plan.finalize_packing(); plan.run_packing()


def build(gs):
    ops = []
    for b0 in range(0, B, gs):
        nb = min(gs, B - b0)
        for r in range(NBLK):
            for k in range(5):
                wp, bp, cin, cout = layers[r * 5 + k]
                n0 = len(ops)
                if k < 4:
                    plan.conv(ops, (R[r], 0), (R[r], 8 + 4 * k), wp, bp, nb, H, W, cin, cout, act=0.2)
                else:
                    plan.conv(ops, (R[r], 0), (R[r + 1], 0), wp, bp, nb, H, W, cin, cout, alpha=0.2, res1=(R[r], 0), beta1=1.0)
                d = ops[n0].u.conv
                es = R[r].esize
                d.x += b0 * d.x_img_stride * es
                d.y += b0 * d.y_img_stride * es
                if d.res1:
                    d.res1 += b0 * d.res1_img_stride * es
    return plan.freeze(ops)


fl = sum(2.0 * B * H * W * 9 * c[2] * c[3] for c in layers)
for gs in groups:
    fr = build(gs)
    plan.run(fr); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); plan.run(fr); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print(f"B{B} {H}x{W} groups of {gs}: {best:8.3f} ms per {NBLK} blocks  {fl / best / 1e9:8.1f} TF/s  check {float(R[NBLK].t.float().abs().mean()):.5f}", flush=True)
