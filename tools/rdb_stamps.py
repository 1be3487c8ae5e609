"""rdb_stamps.py B,H,W: shader-clock stamps of one workgroup of the fused dense-block kernel (a library
built with -DRDB_STAMPS: tools/build_variant.sh stamps "-DRDB_STAMPS"): cycles from kernel start to the
end of input staging, to the end of each layer's K loop and of its epilogue + barrier, per wave."""
import ctypes, importlib, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
os.environ.setdefault("SRX_LIB_PATH", os.path.join(os.path.dirname(os.path.abspath(__file__)), "variants", "libsrx_stamps.so"))
E = importlib.import_module("super-resolution_amd.engine")
G = importlib.import_module("super-resolution_amd.esrgan.generator")
L = importlib.import_module("super-resolution_amd._lib")
dev = torch.device("cuda:0")
B, H, W = [int(t) for t in sys.argv[1].split(",")]
lib = L.lib()
lib.srx_rdb_debug_stamps.argtypes = [ctypes.c_void_p]
E._GenBase.fused_rdb = E._GenBase.fused_rdb_bwd = True
g = G.GeneratorESRGAN(num_blocks=2).to(dev)
params = {n: p.data for n, p in g.named_parameters()}
plan = E.GeneratorPlan(params, B, H, W, torch.bfloat16, num_blocks=2, need_grad=True)
x = torch.rand(B, 3, H, W, device=dev) * 2 - 1
gy = torch.rand(B, 3, 4 * H, 4 * W, device=dev)
buf = torch.zeros(4 * 32, dtype=torch.int64, device=dev)      # the kernel writes [wave][32]
names = ["staged"] + [f"L{l}{s}" for l in range(5) for s in (" loop", " epi")]
for phase in ("forward", "backward"):
    buf.zero_()
    for rep in range(3):
        lib.srx_rdb_debug_stamps(None)
        plan.forward(x)
        if phase == "backward":
            lib.srx_rdb_debug_stamps(buf.data_ptr())
            plan.backward(gy)
        else:
            lib.srx_rdb_debug_stamps(buf.data_ptr())
            plan.forward(x)
        torch.cuda.synchronize()
    lib.srx_rdb_debug_stamps(None)
    t = buf.cpu().view(4, 32)
    print(phase, "(last fused launch, workgroup 8): cycles since kernel start, per wave")
    allnames = names + ([n + " (2nd tile)" for n in names[1:]] if int(t[0, len(names) + 1]) > 0 else [])
    for k, nm in enumerate(allnames):
        print(f"  {nm:18s}", *[f"{int(t[w, k + 1] - t[w, 0]):8d}" for w in range(4)],
              "   delta", *[f"{int(t[w, k + 1] - t[w, k]):7d}" for w in range(4)])
