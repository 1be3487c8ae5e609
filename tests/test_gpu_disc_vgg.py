"""GPU parity of the discriminator, spectral norm, VGG-feature loss and one adversarial
step (drop-in modules through the C ABI) against the reference-generated goldens."""
import pytest
import torch

from helpers import O, TOL, fill_module_, full_grads_bad, golden, pkg, rel_err, rel_l2
from test_gpu_generator import _grads_close

pytestmark = pytest.mark.gpu


def _disc(salt, tdtype):
    D = pkg("esrgan.discriminator")
    d = fill_module_(D.Discriminator(), salt=salt).cuda()
    d.compute_dtype = tdtype
    return d


def test_discriminator_fp32_matches_reference():
    man, arr = golden()
    case = man["cases"]["d"]
    d = _disc(4, torch.float32)
    d.train()
    xr = O.hash_input("d.real", (2, 3, 64, 64)).cuda()
    xf = O.hash_input("d.fake", (2, 3, 64, 64)).cuda().requires_grad_(True)
    lr_ = d(xr)
    lf_ = d(xf)
    assert lr_.shape == (2, 1)
    assert rel_err(lr_, torch.from_numpy(arr["d_logits_real"])) < 1e-3
    assert rel_err(lf_, torch.from_numpy(arr["d_logits_fake"])) < 1e-3
    bce = torch.nn.BCEWithLogitsLoss()
    loss = (bce(lr_, torch.ones_like(lr_)) + bce(lf_, torch.zeros_like(lf_))) / 2
    assert abs(float(loss.detach()) - case["loss_value"]) < 1e-4
    loss.backward()
    # u, v advanced by two power iterations, in place in the module's buffers
    sd = d.state_dict()
    for k in ("features.0.0", "features.7.0", "classifier.1", "classifier.3"):
        for s in ("weight_u", "weight_v"):
            assert rel_err(sd[f"{k}.{s}"], torch.from_numpy(arr[f"d_after2_{k}.{s}"])) < 1e-4
    assert rel_l2(xf.grad, torch.from_numpy(arr["d_gx_fake"])) < 2e-3
    bad = _grads_close(d, case["grads"], 2e-3)
    assert not bad, bad[:5]
    bad = full_grads_bad(((n, p.grad) for n, p in d.named_parameters()), case["full_grads"], arr, 2e-3)
    assert not bad, bad
    # third training forward (the G step's D(fake)) and an eval forward
    l3 = d(xf.detach())
    assert rel_err(l3, torch.from_numpy(arr["d_logits_third"])) < 1e-3
    assert rel_err(d.state_dict()["features.0.0.weight_u"],
                   torch.from_numpy(arr["d_after3_features.0.0.weight_u"])) < 1e-4
    d.eval()
    with torch.no_grad():
        le = d(xr)
    assert rel_err(le, torch.from_numpy(arr["d_logits_eval"])) < 1e-3
    # eval must not advance the power iteration
    assert torch.equal(d.state_dict()["features.0.0.weight_u"].cpu(),
                       sd["features.0.0.weight_u"].cpu())


def test_discriminator_bf16_logits_close():
    man, arr = golden()
    d = _disc(4, torch.bfloat16)
    d.train()
    with torch.no_grad():
        lr_ = d(O.hash_input("d.real", (2, 3, 64, 64)).cuda())
    ref = torch.from_numpy(arr["d_logits_real"])
    assert float((lr_.cpu() - ref).abs().max()) < 2e-2 * max(1.0, float(ref.abs().max()))


def test_discriminator_128_matches_reference():
    man, arr = golden()
    d = _disc(4, torch.float32)
    d.train()
    with torch.no_grad():
        out = d(O.hash_input("d.x128", (2, 3, 128, 128)).cuda())
    assert rel_err(out, torch.from_numpy(arr["d_logits_128"])) < 1e-3


def test_state_dict_layout_matches_reference():
    man, _ = golden()
    D = pkg("esrgan.discriminator")
    G = pkg("esrgan.generator")
    d = D.Discriminator()
    assert [[k, list(v.shape), str(v.dtype)] for k, v in d.state_dict().items()] == man["d_state_dict"]
    assert [k for k, _ in d.named_parameters()] == man["d_named_parameters"]
    assert [k for k, _ in d.named_buffers()] == man["d_named_buffers"]
    meta = d.state_dict()._metadata
    assert {k: v for k, v in meta.items()} == man["d_metadata"]
    g = G.GeneratorESRGAN()
    assert [[k, list(v.shape), str(v.dtype)] for k, v in g.state_dict().items()] == man["g_state_dict"]


@pytest.mark.parametrize("tdtype", [torch.float32, torch.bfloat16])
def test_vgg_features_match_oracle(tdtype):
    """Pinned by the oracle only: torchvision and its pretrained weights are absent."""
    man, arr = golden()
    Lm = pkg("utils.loss")
    vgg_sd = O.make_state_dict(O.vgg19_state_dict_spec(), salt=7, gain=1.4)
    with pytest.warns(UserWarning):
        m = Lm.VGGLoss(device="cuda", allow_synthetic=True)
    m.vgg.load_state_dict(vgg_sd)
    m.compute_dtype = tdtype
    x = O.hash_input("vgg.x", (1, 3, 32, 32))
    pf, _, _, _ = m._get_plans(x.cuda(), False)
    feat = pf.c8_to_nchw((pf.forward_c8(x.cuda(), pack=False), 0), 512)
    ref = torch.from_numpy(arr["vgg_feat"])
    tol = 1e-3 if tdtype == torch.float32 else 3e-2
    assert rel_err(feat, ref) < tol
    # loss + gradient w.r.t. the first argument against the oracle's autograd
    a = O.hash_input("vgg.a", (2, 3, 32, 32)).requires_grad_(True)
    b = O.hash_input("vgg.b", (2, 3, 32, 32))
    ref_loss = O.vgg_loss(a, b, vgg_sd)
    ref_loss.backward()
    ag = a.detach().cuda().requires_grad_(True)
    loss = m(ag, b.cuda())
    assert abs(float(loss.detach()) - float(ref_loss)) / float(ref_loss) < tol
    loss.backward()
    if tdtype == torch.float32:
        assert rel_l2(ag.grad, a.grad) < 2e-2        # sign()/ReLU flips at rounding level
    else:
        cos = torch.nn.functional.cosine_similarity(ag.grad.cpu().flatten(), a.grad.flatten(), dim=0)
        assert float(cos) > 0.9


def test_adversarial_step_fp32_matches_reference():
    """esrgan/train_esrgan.py:87-111 with the drop-in modules: losses and gradients of one
    D step + one G step against the reference's (golden 'adv_step')."""
    man, arr = golden()
    case = man["cases"]["adv_step"]
    G = pkg("esrgan.generator")
    D = pkg("esrgan.discriminator")
    Lm = pkg("utils.loss")
    g = fill_module_(G.GeneratorESRGAN(num_blocks=2), salt=5).cuda()
    d = fill_module_(D.Discriminator(), salt=6).cuda()
    with pytest.warns(UserWarning):
        vgg = Lm.VGGLoss(device="cuda", allow_synthetic=True)
    vgg.vgg.load_state_dict(O.make_state_dict(O.vgg19_state_dict_spec(), salt=7, gain=1.4))
    for m in (g, d, vgg):
        m.compute_dtype = torch.float32
    lr = O.hash_input("adv.lr", (2, 3, 16, 16)).cuda()
    hr = O.hash_input("adv.hr", (2, 3, 64, 64)).cuda()
    bce, l1 = torch.nn.BCEWithLogitsLoss(), torch.nn.L1Loss()
    fake = g(lr)
    assert rel_err(fake.detach().flatten()[::5], torch.from_numpy(arr["adv_fake_s5"])) < 1e-3
    disc_real = d(hr)
    disc_fake = d(fake.detach())
    loss_disc = (bce(disc_real, torch.ones_like(disc_real)) + bce(disc_fake, torch.zeros_like(disc_fake))) / 2
    d.zero_grad(set_to_none=True)
    loss_disc.backward()
    assert abs(float(loss_disc.detach()) - case["loss_disc"]) < 1e-4
    bad = _grads_close(d, case["d_grads"], 3e-3)
    assert not bad, bad[:5]
    bad = full_grads_bad(((n, p.grad) for n, p in d.named_parameters()), case["d_full_grads"], arr, 3e-3)
    assert not bad, bad
    disc_fake_for_gen = d(fake)
    gen_adv = bce(disc_fake_for_gen, torch.ones_like(disc_fake_for_gen))
    gen_l1 = l1(fake, hr)
    gen_vgg = vgg(fake, hr)
    loss_gen = 1.0 * gen_l1 + 5e-3 * gen_adv + 1.0 * gen_vgg
    g.zero_grad(set_to_none=True)
    loss_gen.backward()
    for k, v in (("gen_adv_loss", gen_adv), ("gen_l1_loss", gen_l1), ("gen_vgg_loss", gen_vgg),
                 ("loss_gen", loss_gen)):
        assert abs(float(v.detach()) - case[k]) / abs(case[k]) < 1e-3, k
    bad = _grads_close(g, case["g_grads"], 1e-2)
    assert not bad, bad[:5]
    bad = full_grads_bad(((n, p.grad) for n, p in g.named_parameters()), case["g_full_grads"], arr, 1e-2)
    assert not bad, bad
