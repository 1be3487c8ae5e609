import importlib, sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import O, fill_module_, pkg
import torch.nn.functional as F
D = pkg("esrgan.discriminator")
g = torch.tensor([[0.1182], [0.1182]]).cuda()
d = fill_module_(D.Discriminator(), salt=4).cuda(); d.compute_dtype = torch.float32; d.train()
xf = O.hash_input("d.fake", (2, 3, 64, 64)).cuda().requires_grad_(True)
lf_ = d(xf)
lf_.backward(g)
pb = [p for pool in d._srx_state()["plans"].values() for p in pool][0]
g1 = pb.c8_to_nchw((pb.gA[1], 0), 64)
g0 = pb.c8_to_nchw((pb.gA[0], 0), 3)
W = d.features[0][0].weight_orig.detach() * pb.inv_sigma[0]
exp = F.conv_transpose2d(g1, W, padding=1)
print("g1 max", float(g1.abs().max()), "g0 max", float(g0.abs().max()), "expected g0 max", float(exp.abs().max()))
print("A0 vs x", float((pb.c8_to_nchw((pb.A[0], 0), 3) - xf.detach()).abs().max()))
# check all conv op descriptors of bwd list
import ctypes
for i in range(len(pb.bwd_ops)):
    op = pb.bwd_ops[i]
    if op.kind == 1:
        c = op.u.conv
        print(i, "conv cin", c.cin, "cout", c.cout, "H", c.H, "W", c.W, "ks", c.ksize, "stride", c.stride, "x", hex(c.x or 0), "y", hex(c.y or 0), "w", hex(c.w or 0), "mask", hex(c.mask or 0))
print("gA0 ptr", hex(pb.gA[0].ptr()), "gA1 ptr", hex(pb.gA[1].ptr()))
