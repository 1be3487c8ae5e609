"""Host enqueue time vs GPU time of the adversarial step (is the step bound by the launching thread?)."""
import importlib, sys, time, warnings, torch
sys.path.insert(0, '.')
G = importlib.import_module("super-resolution_amd.esrgan.generator")
Dm = importlib.import_module("super-resolution_amd.esrgan.discriminator")
Lm = importlib.import_module("super-resolution_amd.utils.loss")
T = importlib.import_module("super-resolution_amd.train")
dev = torch.device("cuda:0")
h = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.manual_seed(0)
gen, disc = G.GeneratorESRGAN().to(dev), Dm.Discriminator().to(dev)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    vgg = Lm.VGGLoss(device=dev)
tr = T.AdversarialTrainer(gen, disc, vgg, (16, 3, h, h), torch.bfloat16)
lr = torch.rand(16, 3, h, h, device=dev) * 2 - 1
hr = torch.rand(16, 3, 4 * h, 4 * h, device=dev) * 2 - 1
for _ in range(5): tr.step(lr, hr)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n): tr.step(lr, hr)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
torch.cuda.synchronize(); ta = time.perf_counter(); tr.step(lr, hr); tb = time.perf_counter(); torch.cuda.synchronize(); tc = time.perf_counter()
print(f"  one step from an idle queue: host {1e3 * (tb - ta):.2f} ms, total {1e3 * (tc - ta):.2f} ms")
print(f"adversarial step at {h}x{h}: host enqueue {1e3 * (t1 - t0) / n:.2f} ms/step, total {1e3 * (t2 - t0) / n:.2f} ms/step")
w = T.GeneratorWarmupTrainer(gen, (16, 3, 64, 64), torch.bfloat16)
lr = torch.rand(16, 3, 64, 64, device=dev) * 2 - 1
hr = torch.rand(16, 3, 256, 256, device=dev) * 2 - 1
for _ in range(5): w.step(lr, hr)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n): w.step(lr, hr)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
torch.cuda.synchronize(); ta = time.perf_counter(); w.step(lr, hr); tb = time.perf_counter(); torch.cuda.synchronize(); tc = time.perf_counter()
print(f"  one step from an idle queue: host {1e3 * (tb - ta):.2f} ms, total {1e3 * (tc - ta):.2f} ms")
print(f"warm-up step at 64x64: host enqueue {1e3 * (t1 - t0) / n:.2f} ms/step, total {1e3 * (t2 - t0) / n:.2f} ms/step")
