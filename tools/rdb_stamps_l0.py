"""rdb_stamps_l0.py: per-chunk stamps of the first layer of the fused dense-block forward at 16 x 64 x 64 (a library
built with -DRDB_STAMPS -DRDB_STAMPS_L0); cycles since kernel start, per wave."""
import ctypes, importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SRX_LIB_PATH"] = os.path.join(ROOT, "tools", "variants", "libsrx_stamps.so")
E = importlib.import_module("super-resolution_amd.engine")
G = importlib.import_module("super-resolution_amd.esrgan.generator")
L = importlib.import_module("super-resolution_amd._lib")
dev = torch.device("cuda:0")
lib = L.lib(); lib.srx_rdb_debug_stamps.argtypes = [ctypes.c_void_p]
E._GenBase.fused_rdb = True
g = G.GeneratorESRGAN(num_blocks=2).to(dev)
params = {n: p.data for n, p in g.named_parameters()}
plan = E.GeneratorPlan(params, 16, 64, 64, torch.bfloat16, num_blocks=2, need_grad=True)
x = torch.rand(16, 3, 64, 64, device=dev) * 2 - 1
buf = torch.zeros(128, dtype=torch.int64, device=dev)
for rep in range(3):
    lib.srx_rdb_debug_stamps(None); plan.forward(x)
    lib.srx_rdb_debug_stamps(buf.data_ptr()); plan.forward(x); torch.cuda.synchronize()
lib.srx_rdb_debug_stamps(None)
t = buf.cpu().view(4, 32)
for w in range(4):
    print("wave", w, [int(t[w, k] - t[w, 0]) for k in range(1, 32) if int(t[w, k]) > 0])
