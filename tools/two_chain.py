"""Forward (and fwd+bwd) of the 23-block generator as one chain of B=16 vs two concurrent chains of B=8
on two streams: does the second chain hide the first one's per-launch fixed cost?"""
import ctypes as C, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import pkg
E, L = pkg("engine"), pkg("_lib")
G = pkg("esrgan.generator")
lib = L.lib()
torch.manual_seed(0)
g = G.GeneratorESRGAN(num_blocks=23).cuda()
params = {n: p.data for n, p in g.named_parameters()}
need_grad = len(sys.argv) > 1 and sys.argv[1] == "bwd"
def mk(B):
    return E.GeneratorPlan(params, B, 64, 64, torch.bfloat16, num_blocks=23, need_grad=need_grad)
x16 = torch.rand(16, 3, 64, 64, device="cuda")
gy = torch.rand(16, 3, 256, 256, device="cuda") * 1e-3
def mkstream(prio):
    sp = C.c_void_p(); assert lib.srx_stream_create(prio, C.byref(sp)) == 0
    return torch.cuda.ExternalStream(sp.value)
def run(plans, streams, xs, gys, iters=20):
    def once():
        for p, st, x, gg in zip(plans, streams, xs, gys):
            with torch.cuda.stream(st):
                p.forward_c8(x)
                if need_grad:
                    p.nchw_to_c8(gg, (p.GY, 0)); p.backward_c8()
    for _ in range(3): once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): once()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3
main = torch.cuda.current_stream()
p16 = mk(16)
print("one chain B=16: %.3f ms" % run([p16], [main], [x16], [gy]))
del p16
pa, pb = mk(8), mk(8)
xs, gys = [x16[:8].contiguous(), x16[8:].contiguous()], [gy[:8].contiguous(), gy[8:].contiguous()]
print("two chains B=8 same stream: %.3f ms" % run([pa, pb], [main, main], xs, gys))
s1, s2 = mkstream(0), mkstream(0)
print("two chains B=8, two normal streams: %.3f ms" % run([pa, pb], [s1, s2], xs, gys))
s3 = mkstream(-1)
print("two chains B=8, normal + high: %.3f ms" % run([pa, pb], [s1, s3], xs, gys))
