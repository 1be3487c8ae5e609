import importlib, sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import O, fill_module_, pkg, golden
man, arr = golden()
D = pkg("esrgan.discriminator")
d = fill_module_(D.Discriminator(), salt=4).cuda(); d.compute_dtype = torch.float32; d.train()
xr = O.hash_input("d.real", (2, 3, 64, 64)).cuda()
xf = O.hash_input("d.fake", (2, 3, 64, 64)).cuda().requires_grad_(True)
lr_ = d(xr); lf_ = d(xf)
bce = torch.nn.BCEWithLogitsLoss()
loss = (bce(lr_, torch.ones_like(lr_)) + bce(lf_, torch.zeros_like(lf_))) / 2
lf_.register_hook(lambda g: print("grad wrt lf_", g))
loss.backward()
ref = torch.from_numpy(arr["d_gx_fake"])
print("ours max", float(xf.grad.abs().max()), "ref max", float(ref.abs().max()))
print("ratio", float((xf.grad.cpu() * ref).sum() / (ref * ref).sum()))
