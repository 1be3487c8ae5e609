import importlib, sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import O, fill_module_, pkg
import torch.nn.functional as F
D = pkg("esrgan.discriminator"); C = pkg("chain")
d = fill_module_(D.Discriminator(), salt=4).cuda(); d.compute_dtype = torch.float32; d.train()
x = O.hash_input("d.fake", (2, 3, 64, 64))
sd = {k: v.detach().cpu().clone() for k, v in d.state_dict().items()}
# oracle with intermediate grads
state = {k: v.clone() for k, v in sd.items() if k.endswith(("weight_u", "weight_v"))}
xs = x.clone().requires_grad_(True)
acts = [xs]; h = xs
for i, (_, _, st) in enumerate(O.D_FEATURES):
    n = f"features.{i}.0"
    w, u2, v2 = O.sn_weight(sd[n + ".weight_orig"], state[n + ".weight_u"], state[n + ".weight_v"], True)
    h = O.lrelu(O.conv3x3(h, w, sd[n + ".bias"], st)); h.retain_grad(); acts.append(h)
p = h.mean(dim=(2, 3), keepdim=True); p.retain_grad(); acts.append(p)
w, _, _ = O.sn_weight(sd["classifier.1.weight_orig"], state["classifier.1.weight_u"], state["classifier.1.weight_v"], True)
q = O.lrelu(F.conv2d(p, w, sd["classifier.1.bias"])); q.retain_grad(); acts.append(q)
w, _, _ = O.sn_weight(sd["classifier.3.weight_orig"], state["classifier.3.weight_u"], state["classifier.3.weight_v"], True)
z = F.conv2d(q, w, sd["classifier.3.bias"]); acts.append(z)
gl = torch.tensor([[0.3], [-0.7]])
z.reshape(2, 1).backward(gl)
xg = x.cuda().requires_grad_(True)
out = d(xg); out.backward(gl.cuda()); torch.cuda.synchronize()
plan = [p for pool in d._srx_state()["plans"].values() for p in pool][0]
print("logits", out.detach().cpu().flatten(), z.detach().flatten())
for i in range(len(plan.gA) - 1, -1, -1):
    if plan.gA[i] is None: continue
    b, c, hh, ww = plan.geo[i]
    got = plan.c8_to_nchw((plan.gA[i], 0), c).cpu()
    ref = acts[i].grad
    if ref is None: print(i, "no ref"); continue
    if i in (9, 10):   # batch-as-width layout: [1, c, 1, B] -> [B, c, 1, 1]
        got = got.permute(3, 1, 2, 0)
    # our gA is masked by act'(A[i]) ; oracle's grad is wrt post-activation -> apply mask
    slope = plan.slopes[i]
    if slope != 1.0:
        ref = ref * torch.where(acts[i].detach() > 0, 1.0, slope)
    print(i, plan.geo[i], "relL2 %.3e" % float((got - ref).norm() / ref.norm()), float(got.abs().max()), float(ref.abs().max()))
