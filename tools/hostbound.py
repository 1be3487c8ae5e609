"""Host time to ENQUEUE a training step against the step's duration on the GPU: a step whose enqueue
takes as long as its execution is bound by the host thread, not by the kernels."""
import importlib, os, sys, time, warnings, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
G = importlib.import_module("super-resolution_amd.esrgan.generator")
T = importlib.import_module("super-resolution_amd.train")
dev = torch.device("cuda:0")
adv = len(sys.argv) > 1 and sys.argv[1] == "c3"
B, h = 16, (32 if adv else 64)
if len(sys.argv) > 2:
    h = int(sys.argv[2])
torch.manual_seed(0)
gen = G.GeneratorESRGAN().to(dev)
if adv:
    Dm = importlib.import_module("super-resolution_amd.esrgan.discriminator")
    Lm = importlib.import_module("super-resolution_amd.utils.loss")
    disc = Dm.Discriminator().to(dev)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vgg = Lm.VGGLoss(device=dev, allow_synthetic=True)
    tr = T.AdversarialTrainer(gen, disc, vgg, (B, 3, h, h), torch.bfloat16)
else:
    tr = T.GeneratorWarmupTrainer(gen, (B, 3, h, h), torch.bfloat16)
lr = (torch.rand(B, 3, h, h) * 2 - 1).to(dev)
hr = (torch.rand(B, 3, 4 * h, 4 * h) * 2 - 1).to(dev)
for _ in range(5):
    tr.step(lr, hr)
torch.cuda.synchronize()
one = []
for _ in range(10):                 # queues empty: pure host cost of enqueueing one step
    torch.cuda.synchronize()
    a = time.perf_counter()
    tr.step(lr, hr)
    one.append(time.perf_counter() - a)
torch.cuda.synchronize()
print(f"enqueue of one step into empty queues: min {min(one) * 1e3:.3f} ms, median {sorted(one)[5] * 1e3:.3f} ms")
N = 30
t0 = time.perf_counter()
enq = 0.0
for _ in range(N):
    a = time.perf_counter()
    tr.step(lr, hr)
    enq += time.perf_counter() - a
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"{'c3' if adv else 'c2'} {h}x{h}: enqueue {enq / N * 1e3:.3f} ms/step (loop {t_enq / N * 1e3:.3f}), with final sync {t_all / N * 1e3:.3f} ms/step")
