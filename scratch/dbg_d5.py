import importlib, sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import O, fill_module_, pkg
D = pkg("esrgan.discriminator")
g = torch.tensor([[0.1182], [0.1182]]).cuda()
def run(sync_between, first_fwd, req_first=False):
    d = fill_module_(D.Discriminator(), salt=4).cuda(); d.compute_dtype = torch.float32; d.train()
    xr = O.hash_input("d.real", (2, 3, 64, 64)).cuda()
    xf = O.hash_input("d.fake", (2, 3, 64, 64)).cuda().requires_grad_(True)
    if first_fwd:
        lr_ = d(xr)
    if sync_between: torch.cuda.synchronize()
    lf_ = d(xf)
    lf_.backward(g)
    return float(xf.grad.abs().max())
print("first_fwd=False           ", run(False, False))
print("first_fwd=True  sync=False", run(False, True))
print("first_fwd=True  sync=True ", run(True, True))
print("first_fwd=True  sync=False", run(False, True))
