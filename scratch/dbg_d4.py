import importlib, sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import O, fill_module_, pkg, golden
D = pkg("esrgan.discriminator"); C = pkg("chain")
orig = C.DiscriminatorPlan.backward
def patched(self, g):
    print("plan backward: need_input_grad", self.need_input_grad, "g", g.flatten().tolist(), "stream", torch.cuda.current_stream())
    r = orig(self, g)
    if r is not None: print("  gx max", float(r.abs().max()))
    return r
C.DiscriminatorPlan.backward = patched
d = fill_module_(D.Discriminator(), salt=4).cuda(); d.compute_dtype = torch.float32; d.train()
xr = O.hash_input("d.real", (2, 3, 64, 64)).cuda()
xf = O.hash_input("d.fake", (2, 3, 64, 64)).cuda().requires_grad_(True)
lr_ = d(xr); lf_ = d(xf)
print("fwd stream", torch.cuda.current_stream())
bce = torch.nn.BCEWithLogitsLoss()
loss = (bce(lr_, torch.ones_like(lr_)) + bce(lf_, torch.zeros_like(lf_))) / 2
loss.backward()
print("xf.grad max", float(xf.grad.abs().max()))
plans = [p for pool in d._srx_state()["plans"].values() for p in pool]
pb = [p for p in plans if p.need_input_grad][0]
torch.cuda.synchronize()
for i in range(len(pb.gA) - 1, -1, -1):
    if pb.gA[i] is None: continue
    b, c, hh, ww = pb.geo[i]
    got = pb.c8_to_nchw((pb.gA[i], 0), c)
    print(i, pb.geo[i], "max", float(got.abs().max()), "inv_sigma", float(pb.inv_sigma[min(i, 9)]) if i < 10 else "")
g = torch.tensor([[0.11819113045930862], [0.11824701726436615]]).cuda()
C.DiscriminatorPlan.backward = orig
r1 = pb.backward(g); print("re-run backward: gx max", float(r1.abs().max()))
pb.run_packing()
r2 = pb.backward(g); print("after re-pack: gx max", float(r2.abs().max()))
print("inv_sigma", pb.inv_sigma.tolist())
