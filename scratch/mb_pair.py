import importlib, sys, torch
sys.path.insert(0, '.')
E = importlib.import_module("super-resolution_amd.engine")
dev = torch.device("cuda:0")
def bench(cin_a, mode, pair, B=16, H=64, W=64, reps=100):
    plan = E.Plan(dev, torch.bfloat16)
    bufs = [plan.buf(B, cin_a + 64, H, W, zero=True) for _ in range(4)]
    for b in bufs: b.t.normal_()
    msk = plan.buf(B, 64, H, W); msk.t.normal_()
    wa = torch.randn(32, cin_a, 3, 3, device=dev) * 0.05
    wb = torch.randn(32, cin_a + 32, 3, 3, device=dev) * 0.05
    z = torch.zeros(32, device=dev)
    fwd = mode == "fwd"
    wpa, bpa = plan.packed_layer(32, cin_a, 9, 0, [E.conv_seg(wa)], z.data_ptr() if fwd else 0, with_bias=fwd)
    wpb, bpb = plan.packed_layer(32, cin_a + 32, 9, 0, [E.conv_seg(wb)], z.data_ptr() if fwd else 0, with_bias=fwd)
    plan.finalize_packing(); plan.run_packing()
    x_oct = 0 if fwd else 8
    a_oct, b_oct = (cin_a // 8, cin_a // 8 + 4) if fwd else (4, 0)
    ops = []
    for i in range(reps):
        buf = bufs[i % 4]
        if pair:
            plan.conv_pair(ops, (buf, x_oct), (buf, a_oct), (buf, b_oct), wpa, wpb, bpa, bpb, B, H, W, cin_a,
                           act=0.2 if fwd else 1.0, mask_a=None if fwd else (msk, 4), mask_b=None if fwd else (msk, 0),
                           mask_slope=0.2, a_out_first=not fwd)
        else:
            kw_a = dict(act=0.2) if fwd else dict(mask=(msk, 4), mask_slope=0.2)
            kw_b = dict(act=0.2) if fwd else dict(mask=(msk, 0), mask_slope=0.2)
            plan.conv(ops, (buf, x_oct), (buf, a_oct), wpa, bpa, B, H, W, cin_a, 32, **kw_a)
            plan.conv(ops, (buf, 0) if fwd else (buf, 4), (buf, b_oct), wpb, bpb, B, H, W, cin_a + 32, 32, **kw_b)
    fr = plan.freeze(ops)
    plan.run(fr); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); plan.run(fr); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for mode in ("fwd", "bwd"):
    for cin in (64, 128):
        print(mode, cin, "separate %.2f us   pair %.2f us" % (bench(cin, mode, False), bench(cin, mode, True)))
print("fwd 128 B64 128x128: separate %.1f pair %.1f" % (bench(128, "fwd", False, 64, 128, 128, 20), bench(128, "fwd", True, 64, 128, 128, 20)))
