"""GPU parity of the discriminator, spectral norm, VGG-feature loss and one adversarial
step (drop-in modules through the C ABI) against the reference-generated goldens."""
import pytest
import torch

from helpers import O, TOL, fill_module_, full_grads_bad, golden, pkg, record_observed, rel_err, rel_l2
from test_gpu_generator import _grads_close

pytestmark = pytest.mark.gpu

# fp32 gradients that pass through sign() (L1) and ReLU / LeakyReLU derivatives on tiny fixtures
# (2 x 16x16 LR): one rounding-level sign flip would move a tensor by O(1/pixels).  Observed on MI355X:
# 2.2e-6 relative L2 on every tensor (profiles/r3_observed_parity_errors.json), so the bar is the
# fp32 tolerance of the whole path, 1e-3.
VGG_GRAD_FP32_TOL = 1e-3
ADV_G_GRAD_FP32_TOL = 1e-3


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


def test_discriminator_128_backward_fp32_matches_reference():
    """Config 3's real map sizes (HR 128 x 128: 128, 64, 32, 16, 8 pixels a side,
    esrgan/discriminator.py:18-25): BCE backward through the drop-in module, fp32 MFMA, against the
    reference's whole gradient tensors, input gradient and u / v after the pass."""
    man, arr = golden()
    case = man["cases"]["d128_bwd"]
    d = _disc(4, torch.float32)
    d.train()
    x = O.hash_input("d.x128", (2, 3, 128, 128)).cuda().requires_grad_(True)
    out = d(x)
    assert rel_err(out, torch.from_numpy(arr["d_logits_128"])) < 1e-3
    loss = torch.nn.BCEWithLogitsLoss()(out, torch.ones_like(out))
    assert abs(float(loss.detach()) - case["loss_value"]) < 1e-4
    loss.backward()
    assert rel_l2(x.grad, torch.from_numpy(arr["d128_gx"])) < 2e-3
    bad = _grads_close(d, case["grads"], 2e-3)
    assert not bad, bad[:5]
    bad = full_grads_bad(((n, p.grad) for n, p in d.named_parameters()), case["full_grads"], arr, 2e-3)
    assert not bad, bad
    sd = d.state_dict()
    for k in ("features.0.0", "features.6.0", "classifier.3"):
        for s in ("weight_u", "weight_v"):
            assert rel_err(sd[f"{k}.{s}"], torch.from_numpy(arr[f"d128_after1_{k}.{s}"])) < 1e-4


def test_discriminator_128_backward_bf16_kernels_meet_tolerance():
    """The bf16 gradient kernels of the discriminator chain at 128 x 128 (stride-1 and stride-2
    data gradients, weight gradients, spectral-norm backward) against the REFERENCE's gradients.
    As for the generator, the activation-derivative masks are those of an fp32 forward: its saved
    activations (rounded to bf16, signs unchanged) are placed in the bf16 plan's buffers, so a
    rounding-level sign flip of a near-zero activation (an O(1) change of one incoherent term at any
    bf16 forward) does not enter; every whole gradient tensor must then meet 2e-2.  With the masks
    of its own bf16 forward the direction of the whole gradient is kept (cosine)."""
    man, arr = golden()
    case = man["cases"]["d128_bwd"]
    CH = pkg("chain")
    D = pkg("esrgan.discriminator")
    d = fill_module_(D.Discriminator(), salt=4).cuda()
    params = {n: p.data for n, p in d.named_parameters()}
    x = O.hash_input("d.x128", (2, 3, 128, 128)).cuda()

    def plan_for(tdtype):
        bufs = {n: b.clone() for n, b in d.named_buffers()}      # each plan advances its own u, v
        return CH.DiscriminatorPlan(params, bufs, 2, 128, 128, tdtype, need_param_grad=True, need_input_grad=True)

    p32, p16 = plan_for(torch.float32), plan_for(torch.bfloat16)
    logits = p32.forward_logits(x, training=True)
    g = (torch.sigmoid(logits) - 1.0) / logits.numel()            # d BCE(logits, 1) / d logits
    p32.backward(g)
    l16 = p16.forward_logits(x, training=True)
    ref_logits = torch.from_numpy(arr["d_logits_128"])
    assert float((l16.cpu() - ref_logits).abs().max()) < 2e-2 * max(1.0, float(ref_logits.abs().max()))
    gx_own = p16.backward(g).clone()
    own = torch.cat([p16.grads[n].flatten() for n in p16.param_names]).double()
    for a, b in zip(p16.A, p32.A):
        a.t.copy_(b.t)
    gx = p16.backward(g)
    torch.cuda.synchronize()
    assert rel_l2(gx, torch.from_numpy(arr["d128_gx"])) < 2e-2
    bad = full_grads_bad(p16.grads.items(), case["full_grads"], arr, 2e-2)
    assert not bad, bad
    ref = torch.cat([p32.grads[n].flatten() for n in p16.param_names]).double()
    assert float(torch.dot(own, ref) / (own.norm() * ref.norm())) > 0.97
    assert float(torch.nn.functional.cosine_similarity(gx_own.flatten().double().cpu(),
                                                       torch.from_numpy(arr["d128_gx"]).flatten().double(), dim=0)) > 0.97


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
        e = rel_l2(ag.grad, a.grad)
        record_observed("vgg_input_grad_fp32", rel_l2=e)
        assert e < VGG_GRAD_FP32_TOL                 # sign()/ReLU flips at rounding level
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
    worst_norm = max(abs(float(p.grad.double().norm()) - case["g_grads"][n]["norm"]) / max(case["g_grads"][n]["norm"], 1e-12)
                     for n, p in g.named_parameters())
    worst_l2 = max(rel_l2(dict(g.named_parameters())[n].grad.float().cpu().flatten()[::e["step"]],
                          torch.from_numpy(arr[e["array"]]).flatten()) for n, e in case["g_full_grads"].items())
    record_observed("adv_step_g_grads_fp32", worst_norm_err=worst_norm, worst_full_rel_l2=worst_l2)
    bad = _grads_close(g, case["g_grads"], ADV_G_GRAD_FP32_TOL)
    assert not bad, bad[:5]
    bad = full_grads_bad(((n, p.grad) for n, p in g.named_parameters()), case["g_full_grads"], arr, ADV_G_GRAD_FP32_TOL)
    assert not bad, bad
