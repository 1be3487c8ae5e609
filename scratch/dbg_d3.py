import importlib, sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import O, fill_module_, pkg, golden
D = pkg("esrgan.discriminator")
def oracle_grad(sd, state, x, gl):
    xs = x.clone().requires_grad_(True)
    out = O.discriminator_forward(xs, sd, True, state)
    out.backward(gl)
    return xs.grad
for scen in ("single", "two_fwd"):
    d = fill_module_(D.Discriminator(), salt=4).cuda(); d.compute_dtype = torch.float32; d.train()
    sd = {k: v.detach().cpu().clone() for k, v in d.state_dict().items()}
    state = {k: v.clone() for k, v in sd.items() if k.endswith(("weight_u", "weight_v"))}
    xr = O.hash_input("d.real", (2, 3, 64, 64)); xf = O.hash_input("d.fake", (2, 3, 64, 64))
    gl = torch.tensor([[0.1182], [0.1182]])
    if scen == "two_fwd":
        lr_ = d(xr.cuda()); O.discriminator_forward(xr, sd, True, state)
    xg = xf.cuda().requires_grad_(True)
    lf_ = d(xg)
    ref = oracle_grad(sd, state, xf, gl)
    lf_.backward(gl.cuda())
    print(scen, "ours max", float(xg.grad.abs().max()), "ref max", float(ref.abs().max()), "relL2", float((xg.grad.cpu() - ref).norm() / ref.norm()))
    if scen == "two_fwd":
        (lr_.sum() * 0.1).backward()
        print("after second backward ok")
