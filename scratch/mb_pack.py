import importlib, sys, torch
sys.path.insert(0, '.')
L = importlib.import_module("super-resolution_amd._lib")
if len(sys.argv) > 1 and sys.argv[1] != "base":
    L.LIB_PATH = "scratch/libsrx_%s.so" % sys.argv[1]
E = importlib.import_module("super-resolution_amd.engine")
G = importlib.import_module("super-resolution_amd.esrgan.generator")
dev = torch.device("cuda:0")
gen = G.GeneratorESRGAN().to(dev)
params = {n: p.data for n, p in gen.named_parameters()}
plan = E.GeneratorPlan(params, 16, 64, 64, torch.bfloat16, need_grad=True, need_input_grad=False)
ls = plan.pack_layers
units = [l.nblk * l.nchunk for l in ls]
print("layers", len(ls), "units", sum(units), "max", max(units), "elems", sum(u * l.taps * 512 for u, l in zip(units, ls)),
      "segs", len(plan.pack_segs), "max_elems", plan.pack_max_elems)
from collections import Counter
print(Counter(units).most_common(12))
plan.run_packing(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): plan.run_packing()
e1.record(); torch.cuda.synchronize()
print("pack %.1f us" % (e0.elapsed_time(e1) * 1e3 / 20))
