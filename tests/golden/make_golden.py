"""Generate the golden fixtures in this directory from the REAL reference.

Run only in the container that has /root/reference (it is absent on the GPU
box):  ``python tests/golden/make_golden.py``.  It imports the reference's
``esrgan.generator`` / ``esrgan.discriminator`` modules on the CPU, fills their
parameters with the deterministic hash filler of ``oracle/esrgan_oracle.py``
(so no weights need committing), runs them, and stores inputs' recipes and the
expected outputs / gradients as small ``.npz`` / ``.json`` / ``.pth`` files.

The fixtures are data only: no text of the reference travels.
``utils/loss.py`` (VGGLoss) cannot be imported here (torchvision is not
installed and its pretrained weights are a network fetch), so the perceptual
loss fixtures come from the oracle's restatement with synthetic weights and are
marked ``pinned_by: "oracle"`` in the manifest.
"""
import json
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from oracle import esrgan_oracle as O   # noqa: E402

from esrgan.generator import (GeneratorESRGAN, RRDB, ResidualDenseBlock,  # noqa: E402
                              UpsampleBlock)
from esrgan.discriminator import Discriminator  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)


def fill_module_(m, salt=0, gain=1.0):
    sd = m.state_dict()
    O.fill_state_dict_(sd, salt, gain)      # state_dict tensors alias the module's storage
    return m


def grad_summary(named_params):
    """Per-parameter: L2 norm (fp64), sum (fp64), first 8 flat elements."""
    out = OrderedDict()
    for k, p in named_params:
        g = p.grad.detach().double().flatten()
        out[k] = {"norm": float(g.norm()), "sum": float(g.sum()),
                  "head": [float(v) for v in g[:8]]}
    return out


def full_grads(case_name, named_params, names, steps=None):
    """Whole gradient tensors of the named parameters (every `steps[name]`-th flat element for
    the few large ones), stored as `<case>_grad::<name>`: element-wise evidence that a
    permutation of rows or taps inside a tensor cannot pass."""
    steps = steps or {}
    P = dict(named_params)
    index = {}
    for n in names:
        g = P[n].grad.detach().float()
        st = steps.get(n, 1)
        key = f"{case_name}_grad::{n}"
        arrays[key] = (g.flatten()[::st] if st > 1 else g).numpy().copy()
        index[n] = {"array": key, "step": st}
    return index


def sample(t, step):
    return t.detach().flatten()[::step].float().numpy().copy()


manifest = {"filler": "oracle.esrgan_oracle.hash_uniform / fill_state_dict_", "cases": {}}
arrays = {}

BLOCK_FULL = {"rdb": ["conv1.weight", "conv3.bias", "conv5.weight"],
              "rrdb": ["rdb1.conv2.weight", "rdb2.conv1.weight", "rdb3.conv5.weight", "rdb3.conv5.bias"],
              "upsample": ["conv.weight", "conv.bias"]}


def case(name, **kw):
    manifest["cases"][name] = kw


# ---------------------------------------------------------------- pixel shuffle
x = torch.arange(2 * 8 * 3 * 5, dtype=torch.float32).reshape(2, 8, 3, 5)
ps = torch.nn.PixelShuffle(2)(x)
assert torch.equal(ps, O.pixel_shuffle2(x))
arrays["pixel_shuffle_out"] = ps.numpy().astype(np.int32)
case("pixel_shuffle", input="arange(2*8*3*5).reshape(2,8,3,5)", out="pixel_shuffle_out",
     pinned_by="reference (nn.PixelShuffle(2), esrgan/generator.py:44)")

# ---------------------------------------------------------------- RDB / RRDB / Upsample
for name, mod, shape in (("rdb", ResidualDenseBlock(64, 32), (2, 64, 12, 20)),
                         ("rrdb", RRDB(64), (2, 64, 12, 20)),
                         ("upsample", UpsampleBlock(64), (2, 64, 6, 10))):
    fill_module_(mod, salt=1)
    xin = O.hash_input(name + ".x", shape).requires_grad_(True)
    y = mod(xin)
    gout = O.hash_input(name + ".gy", y.shape)
    (y * gout).sum().backward()
    sd = {k: v.detach() for k, v in mod.state_dict().items()}
    fn = {"rdb": O.rdb_forward, "rrdb": O.rrdb_forward, "upsample": O.upsample_forward}[name]
    err = (fn(xin.detach(), sd) - y.detach()).abs().max().item()
    assert err == 0.0, (name, err)
    arrays[name + "_y"] = y.detach().numpy()
    arrays[name + "_gx"] = xin.grad.numpy()
    case(name, x=f"hash_input('{name}.x', {list(shape)})", gy=f"hash_input('{name}.gy', y.shape)",
         loss="sum(y*gy)", salt=1, y=name + "_y", gx=name + "_gx",
         grads=grad_summary(mod.named_parameters()),
         full_grads=full_grads(name, mod.named_parameters(), BLOCK_FULL[name]), pinned_by="reference")

# ---------------------------------------------------------------- generator, 1 block, small
g1 = fill_module_(GeneratorESRGAN(num_blocks=1), salt=2)
lr = O.hash_input("g1.lr", (2, 3, 16, 24)).requires_grad_(True)
hr = O.hash_input("g1.hr", (2, 3, 64, 96))
out = g1(lr)
loss = torch.nn.L1Loss()(out, hr)
loss.backward()
sd1 = {k: v.detach() for k, v in g1.state_dict().items()}
assert (O.generator_forward(lr.detach(), sd1) - out.detach()).abs().max().item() == 0.0
arrays["g1_out"] = out.detach().numpy()
arrays["g1_glr"] = lr.grad.numpy()
case("g1", num_blocks=1, lr="hash_input('g1.lr',[2,3,16,24])", hr="hash_input('g1.hr',[2,3,64,96])",
     salt=2, loss="L1Loss(out,hr)", loss_value=float(loss), out="g1_out", glr="g1_glr",
     grads=grad_summary(g1.named_parameters()),
     full_grads=full_grads("g1", g1.named_parameters(),
                           ["conv_first.weight", "conv_first.bias", "body.0.rdb1.conv1.weight",
                            "body.0.rdb2.conv4.weight", "body.0.rdb3.conv5.weight", "conv_body.weight",
                            "upsample.0.conv.weight", "upsample.1.conv.weight", "upsample.1.conv.bias",
                            "conv_hr.weight", "conv_last.weight", "conv_last.bias"]),
     pinned_by="reference")

# ---------------------------------------------------------------- generator, 23 blocks, config 1
g23 = fill_module_(GeneratorESRGAN(), salt=3)
lr = O.hash_input("g23.lr", (4, 3, 32, 32))
hr = O.hash_input("g23.hr", (4, 3, 128, 128))
out = g23(lr)
loss = torch.nn.L1Loss()(out, hr)
loss.backward()
sd23 = {k: v.detach() for k, v in g23.state_dict().items()}
with torch.no_grad():
    err = (O.generator_forward(lr, sd23) - out).abs().max().item()
assert err == 0.0, err
arrays["g23_out_s7"] = sample(out, 7)
case("g23", num_blocks=23, lr="hash_input('g23.lr',[4,3,32,32])", hr="hash_input('g23.hr',[4,3,128,128])",
     salt=3, loss="L1Loss(out,hr)", loss_value=float(loss), out_sample="g23_out_s7", sample_step=7,
     out_sum=float(out.double().sum()), out_abs_sum=float(out.double().abs().sum()),
     grads=grad_summary(g23.named_parameters()),
     full_grads=full_grads("g23", g23.named_parameters(),
                           ["conv_first.weight", "body.0.rdb1.conv1.weight", "body.11.rdb2.conv3.weight",
                            "body.22.rdb3.conv5.weight", "body.22.rdb3.conv5.bias", "conv_body.weight",
                            "upsample.1.conv.weight", "conv_last.weight"],
                           steps={"upsample.1.conv.weight": 4}),
     pinned_by="reference")
manifest["g_state_dict"] = [[k, list(v.shape), str(v.dtype)] for k, v in g23.state_dict().items()]
manifest["g_num_params"] = sum(p.numel() for p in g23.parameters())
spec = O.generator_state_dict_spec(23)
assert [(k, tuple(v.shape)) for k, v in g23.state_dict().items()] == spec
del g23, sd23

# ---------------------------------------------------------------- discriminator
torch.manual_seed(0)
d = Discriminator()
# constructor RNG pin: u/v as drawn by spectral_norm at construction under manual_seed(0)
arrays["d_ctor_u0"] = d.features[0][0].weight_u.detach().numpy().copy()
arrays["d_ctor_v_last"] = d.classifier[3].weight_v.detach().numpy().copy()
arrays["d_ctor_w0_head"] = d.features[0][0].weight_orig.detach().flatten()[:16].numpy().copy()
fill_module_(d, salt=4)
dsd0 = {k: v.detach().clone() for k, v in d.state_dict().items()}
assert [(k, tuple(v.shape)) for k, v in dsd0.items()] == O.discriminator_state_dict_spec()
meta = d.state_dict()._metadata
manifest["d_state_dict"] = [[k, list(v.shape), str(v.dtype)] for k, v in dsd0.items()]
manifest["d_metadata"] = {k: v for k, v in meta.items()}
manifest["d_named_parameters"] = [k for k, _ in d.named_parameters()]
manifest["d_named_buffers"] = [k for k, _ in d.named_buffers()]
manifest["d_num_params"] = sum(p.numel() for p in d.parameters())

state = {k: v.clone() for k, v in dsd0.items() if k.endswith(("weight_u", "weight_v"))}
xr = O.hash_input("d.real", (2, 3, 64, 64))
xf = O.hash_input("d.fake", (2, 3, 64, 64)).requires_grad_(True)
d.train()
lr_ = d(xr)
lf_ = d(xf)
bce = torch.nn.BCEWithLogitsLoss()
loss_d = (bce(lr_, torch.ones_like(lr_)) + bce(lf_, torch.zeros_like(lf_))) / 2
loss_d.backward()
o_lr = O.discriminator_forward(xr, dsd0, True, state)
o_lf = O.discriminator_forward(xf.detach(), dsd0, True, state)
assert (o_lr - lr_.detach()).abs().max().item() <= 1e-6 * lr_.abs().max().item()
assert (o_lf - lf_.detach()).abs().max().item() <= 1e-6 * lf_.abs().max().item()
arrays["d_logits_real"] = lr_.detach().numpy()
arrays["d_logits_fake"] = lf_.detach().numpy()
arrays["d_gx_fake"] = xf.grad.numpy()
for k in ("features.0.0", "features.7.0", "classifier.1", "classifier.3"):
    for s in ("weight_u", "weight_v"):
        ref = d.state_dict()[f"{k}.{s}"]
        assert (ref - state[f"{k}.{s}"]).abs().max().item() < 1e-6
        arrays[f"d_after2_{k}.{s}"] = ref.numpy().copy()
dgr = grad_summary(d.named_parameters())
dfull = full_grads("d", d.named_parameters(),
                   ["features.0.0.weight_orig", "features.0.0.bias", "features.3.0.weight_orig",
                    "features.6.0.weight_orig", "classifier.1.weight_orig", "classifier.1.bias",
                    "classifier.3.weight_orig"],
                   steps={"features.6.0.weight_orig": 16, "classifier.1.weight_orig": 8})
# third train-mode forward (G step) and an eval forward
l3 = d(xf.detach())
arrays["d_logits_third"] = l3.detach().numpy()
arrays["d_after3_features.0.0.weight_u"] = d.state_dict()["features.0.0.weight_u"].numpy().copy()
d.eval()
le = d(xr)
arrays["d_logits_eval"] = le.detach().numpy()
d.train()
case("d", salt=4, real="hash_input('d.real',[2,3,64,64])", fake="hash_input('d.fake',[2,3,64,64])",
     loss="(BCE(D(real),1)+BCE(D(fake),0))/2", loss_value=float(loss_d),
     logits_real="d_logits_real", logits_fake="d_logits_fake", gx_fake="d_gx_fake",
     logits_third="d_logits_third", logits_eval="d_logits_eval",
     grads=dgr, full_grads=dfull, pinned_by="reference")

# 128x128 logits (reference default HR crop, config.py:40)
fill_module_(d, salt=4)
x128 = O.hash_input("d.x128", (2, 3, 128, 128))
arrays["d_logits_128"] = d(x128).detach().numpy()
case("d128", salt=4, x="hash_input('d.x128',[2,3,128,128])", logits="d_logits_128",
     note="first train-mode forward from the salt-4 state", pinned_by="reference")

# 128x128 with backward (config 3's real map sizes: 128, 64, 32, 16, 8 pixels a side,
# esrgan/discriminator.py:18-25): BCE against ones, whole gradient tensors, input gradient, u / v after
fill_module_(d, salt=4)
d.zero_grad(set_to_none=True)
x128g = O.hash_input("d.x128", (2, 3, 128, 128)).requires_grad_(True)
l128 = d(x128g)
assert torch.equal(l128.detach(), torch.from_numpy(arrays["d_logits_128"]))
loss128 = torch.nn.BCEWithLogitsLoss()(l128, torch.ones_like(l128))
loss128.backward()
arrays["d128_gx"] = x128g.grad.numpy().copy()
for k in ("features.0.0", "features.6.0", "classifier.3"):
    for s in ("weight_u", "weight_v"):
        arrays[f"d128_after1_{k}.{s}"] = d.state_dict()[f"{k}.{s}"].numpy().copy()
case("d128_bwd", salt=4, x="hash_input('d.x128',[2,3,128,128])", loss="BCE(D(x),1)", loss_value=float(loss128),
     gx="d128_gx", grads=grad_summary(d.named_parameters()),
     full_grads=full_grads("d128", d.named_parameters(),
                           ["features.0.0.weight_orig", "features.0.0.bias", "features.3.0.weight_orig",
                            "features.6.0.weight_orig", "classifier.1.weight_orig", "classifier.3.weight_orig"],
                           steps={"features.6.0.weight_orig": 16, "classifier.1.weight_orig": 8}),
     note="first train-mode forward from the salt-4 state, then backward", pinned_by="reference")
d.zero_grad(set_to_none=True)

# ---------------------------------------------------------------- SN backward identity (fp64)
w = O.hash_uniform("sn.w", (6, 4, 3, 3)).double().requires_grad_(True)
u = torch.nn.functional.normalize(O.hash_uniform("sn.u", (6,)).double(), dim=0)
v = torch.nn.functional.normalize(O.hash_uniform("sn.v", (36,)).double(), dim=0)
wsn, u2, v2 = O.sn_weight(w, u, v, True)
gw = O.hash_uniform("sn.g", (6, 4, 3, 3)).double()
(wsn * gw).sum().backward()
sigma = torch.dot(u2, w.detach().reshape(6, -1) @ v2)
ident = gw / sigma - ((gw * w.detach()).sum() / sigma ** 2) * torch.outer(u2, v2).reshape(6, 4, 3, 3)
assert (ident - w.grad).abs().max().item() < 1e-12
case("sn_backward_identity", note="dW_orig = dW_sn/sigma - (<dW_sn,W_orig>/sigma^2) u v^T verified in fp64",
     pinned_by="reference (torch.nn.utils.spectral_norm semantics)")

# ---------------------------------------------------------------- full adversarial step
g = fill_module_(GeneratorESRGAN(num_blocks=2), salt=5)
d = fill_module_(Discriminator(), salt=6)
vgg_sd = O.make_state_dict(O.vgg19_state_dict_spec(), salt=7, gain=1.4)
lr = O.hash_input("adv.lr", (2, 3, 16, 16))
hr = O.hash_input("adv.hr", (2, 3, 64, 64))
bce, l1 = torch.nn.BCEWithLogitsLoss(), torch.nn.L1Loss()
fake = g(lr)
disc_real = d(hr)
disc_fake = d(fake.detach())
loss_disc = (bce(disc_real, torch.ones_like(disc_real)) + bce(disc_fake, torch.zeros_like(disc_fake))) / 2
d.zero_grad(set_to_none=True)
loss_disc.backward()
d_grads = grad_summary(d.named_parameters())
d_full = full_grads("adv_d", d.named_parameters(), ["features.0.0.weight_orig", "features.2.0.weight_orig",
                                                    "classifier.3.weight_orig"])
# (no optimizer step between the phases: the fixture pins the loss graph, not Adam)
disc_fake_for_gen = d(fake)
gen_adv = bce(disc_fake_for_gen, torch.ones_like(disc_fake_for_gen))
gen_l1 = l1(fake, hr)
gen_vgg = O.vgg_loss(fake, hr, vgg_sd)
loss_gen = 1.0 * gen_l1 + 5e-3 * gen_adv + 1.0 * gen_vgg
g.zero_grad(set_to_none=True)
loss_gen.backward()
g_grads = grad_summary(g.named_parameters())
g_full = full_grads("adv_g", g.named_parameters(), ["conv_first.weight", "body.1.rdb3.conv5.weight",
                                                    "upsample.0.conv.weight", "conv_last.weight"])
arrays["adv_fake_s5"] = sample(fake, 5)
case("adv_step", g_blocks=2, g_salt=5, d_salt=6, vgg_salt=7, vgg_gain=1.4,
     lr="hash_input('adv.lr',[2,3,16,16])", hr="hash_input('adv.hr',[2,3,64,64])",
     lambdas=[1.0, 5e-3, 1.0],
     loss_disc=float(loss_disc), gen_adv_loss=float(gen_adv), gen_l1_loss=float(gen_l1),
     gen_vgg_loss=float(gen_vgg), loss_gen=float(loss_gen), fake_sample="adv_fake_s5", sample_step=5,
     d_grads=d_grads, g_grads=g_grads, d_full_grads=d_full, g_full_grads=g_full,
     pinned_by="reference G and D; VGG features by the oracle (synthetic weights)")

# oracle VGG-only fixture
xv = O.hash_input("vgg.x", (1, 3, 32, 32))
fv = O.vgg19_features36(xv, vgg_sd)
arrays["vgg_feat"] = fv.numpy()
case("vgg_features", salt=7, gain=1.4, x="hash_input('vgg.x',[1,3,32,32])", feat="vgg_feat",
     pinned_by="oracle (parity unpinned by the reference: torchvision + pretrained weights absent)")

# ---------------------------------------------------------------- reference-format checkpoints (tiny)
gs = fill_module_(GeneratorESRGAN(num_features=16, num_blocks=1), salt=8)
torch.save(gs.state_dict(), os.path.join(HERE, "ref_generator_nf16_nb1.pth"))      # train_esrgan.py:193
xs = O.hash_input("ckpt.x", (1, 3, 8, 8))
arrays["ckpt_g_out"] = gs(xs).detach().numpy()
case("checkpoint_g", file="ref_generator_nf16_nb1.pth", num_features=16, num_blocks=1,
     x="hash_input('ckpt.x',[1,3,8,8])", out="ckpt_g_out", salt=8, pinned_by="reference")

np.savez_compressed(os.path.join(HERE, "golden.npz"), **arrays)
with open(os.path.join(HERE, "golden.json"), "w") as f:
    json.dump(manifest, f, indent=1)
print("wrote", len(arrays), "arrays;", {k: v.shape for k, v in arrays.items()})
