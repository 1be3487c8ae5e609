"""c3_overlap_probe.py: where inside the adversarial step (config 3) the side stream's HR-only work (D(hr) forward +
backward, VGG(hr)) runs relative to the generator's forward pass on the main stream, measured with events in an
un-profiled run (a kernel trace slows the host thread enough to change the picture)."""
import importlib, os, sys, warnings, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
G = importlib.import_module("super-resolution_amd.esrgan.generator")
T = importlib.import_module("super-resolution_amd.train")
Dm = importlib.import_module("super-resolution_amd.esrgan.discriminator")
Lm = importlib.import_module("super-resolution_amd.utils.loss")
dev = torch.device("cuda:0")
B, h = 16, int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.manual_seed(0)
gen, disc = G.GeneratorESRGAN().to(dev), Dm.Discriminator().to(dev)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    vgg = Lm.VGGLoss(device=dev, allow_synthetic=True)
tr = T.AdversarialTrainer(gen, disc, vgg, (B, 3, h, h), torch.bfloat16)
lr = (torch.rand(B, 3, h, h) * 2 - 1).to(dev)
hr = (torch.rand(B, 3, 4 * h, 4 * h) * 2 - 1).to(dev)
ev = {k: [] for k in ("step0", "gfwd_end", "real0", "real1", "step1", "phase1_end", "dstep_end", "vgg0", "vgg1", "gbwd0")}
def mark(k):
    e = torch.cuda.Event(enable_timing=True); e.record(); ev[k].append(e)
orig_fwd, orig_d = tr.gplan.forward_c8, tr._d_pass
def fwd(x):
    y = orig_fwd(x); mark("gfwd_end"); return y
def d_pass(plan, x, fake, target):
    real = plan is tr.d_real
    if real: mark("real0")
    orig_d(plan, x, fake, target)
    if real: mark("real1")
tr.gplan.forward_c8, tr._d_pass = fwd, d_pass
# phase boundaries of the step on the main stream (the trainer brackets its phases with trace.push / pop)
pops = [0]
orig_pop, orig_vgg, orig_l1 = T.trace.pop, tr._vgg_branch, None
def pop():
    orig_pop()
    k = ("phase1_end", "dstep_end", None)[pops[0] % 3]
    pops[0] += 1
    if k: mark(k)
def vgg_branch(*a):
    mark("vgg0"); orig_vgg(*a); mark("vgg1")
T.trace.pop, tr._vgg_branch = pop, vgg_branch
orig_bwd = T._backward_with_exchange
def bwd(gp, sync):
    mark("gbwd0"); return orig_bwd(gp, sync)
T._backward_with_exchange = bwd
for i in range(25):
    mark("step0"); tr.step(lr, hr); mark("step1")
torch.cuda.synchronize()
def med(a, b):
    v = sorted(x.elapsed_time(y) for x, y in list(zip(ev[a], ev[b]))[8:])
    return v[len(v) // 2]
print(f"step {med('step0', 'step1'):.3f} ms; generator forward ends at {med('step0', 'gfwd_end'):.3f}; "
      f"D(hr) pass on the side stream from {med('step0', 'real0'):.3f} to {med('step0', 'real1'):.3f}")
print(f"main stream: phase 1 (G forward, wait for the HR-only work) ends {med('step0', 'phase1_end'):.3f}; discriminator step ends "
      f"{med('step0', 'dstep_end'):.3f}; generator backward starts {med('step0', 'gbwd0'):.3f}; "
      f"perceptual branch on the side stream from {med('step0', 'vgg0'):.3f} to {med('step0', 'vgg1'):.3f}")
