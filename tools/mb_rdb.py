"""mb_rdb.py B,H,W[,blocks,reps] ... : the generator trunk's forward (dense blocks only), stand-alone, as
fused launches (srx_rdb_forward) and as five convolution launches per block: microseconds per block."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
E = importlib.import_module("super-resolution_amd.engine")
G = importlib.import_module("super-resolution_amd.esrgan.generator")
L = importlib.import_module("super-resolution_amd._lib")
dev = torch.device("cuda:0")


def bench(B, H, W, blocks=6, reps=5, need_grad=1):
    torch.manual_seed(0)
    g = G.GeneratorESRGAN(num_blocks=blocks).to(dev)
    params = {n: p.data for n, p in g.named_parameters()}
    x = torch.rand(B, 3, H, W, device=dev) * 2 - 1
    for fused in (False, True):
        E._GenBase.fused_rdb = fused
        plan = E.GeneratorPlan(params, B, H, W, torch.bfloat16, num_blocks=blocks, need_grad=bool(need_grad))
        plan.forward(x)
        n = len(plan.fwd_ops)
        trunk = [i for i in range(n) if (plan.fwd_ops[i].kind == L.OP_RDB_FWD) or
                 (plan.fwd_ops[i].kind == L.OP_CONV and plan.fwd_ops[i].u.conv.H == H and 1 <= i < n - 5)]
        first, last = trunk[0], trunk[-1]
        if not fused:
            last = first + 15 * blocks - 1
        cnt = last - first + 1
        best = 1e9
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = plan.lib.srx_run_ops(E.C.addressof(plan.fwd_ops[first]), cnt, E.stream_ptr(dev))
            e1.record(); torch.cuda.synchronize()
            assert rc == 0
            best = min(best, e0.elapsed_time(e1) * 1e3 / (3 * blocks))
        fl = 2.0 * B * H * W * 239616
        print(f"B{B} {H}x{W} {'fused' if fused else 'five launches'}: {best:8.2f} us per block ({cnt} launches)  "
              f"{fl / best / 1e6:8.1f} TF/s", flush=True)
    E._GenBase.fused_rdb = None


for a in sys.argv[1:]:
    bench(*[int(t) for t in a.split(",")])
