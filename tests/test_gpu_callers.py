"""The call patterns of the reference's training and inference loops on the drop-in modules:
forward under torch.autocast (esrgan/train_esrgan.py:87,101; inference.py:153), backward through
GradScaler-scaled losses with scaler.step / update (:96-98,109-111), a .half() model in eval mode
(inference.py:116-118).  Each must run, stay finite, and give what the explicit bf16 compute path
gives (under autocast or with half parameters the modules compute on bf16 MFMA)."""
import copy

import pytest
import torch

from helpers import O, fill_module_, pkg, rel_err, rel_l2

pytestmark = pytest.mark.gpu


def _gen(blocks=1, salt=5):
    G = pkg("esrgan.generator")
    return fill_module_(G.GeneratorESRGAN(num_blocks=blocks), salt=salt).cuda()


def _disc(salt=6):
    D = pkg("esrgan.discriminator")
    return fill_module_(D.Discriminator(), salt=salt).cuda()


def test_generator_forward_under_autocast_equals_bf16_compute():
    g = _gen()
    x = O.hash_input("ac.x", (2, 3, 16, 24)).cuda()
    with torch.no_grad():
        g.compute_dtype = torch.bfloat16
        want = g(x)
        g.compute_dtype = None
        plain = g(x)                                   # fp32 MFMA without autocast
        with torch.autocast("cuda", dtype=torch.float16):
            got = g(x)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            got_bf = g(x)
    assert got.dtype == torch.float32 and torch.isfinite(got).all()
    assert torch.equal(got, want) and torch.equal(got_bf, want)
    assert not torch.equal(plain, want) and rel_err(want, plain) < 2e-2


def test_train_step_with_autocast_and_gradscaler_as_the_reference_loop():
    """esrgan/train_esrgan.py:87-111 verbatim in structure: autocast forward passes, scaled
    backward, scaler.step, scaler.update for D then G — against the same step computed with
    compute_dtype = bf16 and no scaler."""
    def one_step(use_amp):
        g, d = _gen(blocks=1, salt=5), _disc(6)
        if not use_amp:
            g.compute_dtype = d.compute_dtype = torch.bfloat16
        opt_g = torch.optim.Adam(g.parameters(), lr=2e-4, betas=(0.9, 0.999))
        opt_d = torch.optim.Adam(d.parameters(), lr=2e-4, betas=(0.9, 0.999))
        scaler_g = torch.amp.GradScaler("cuda", enabled=use_amp)
        scaler_d = torch.amp.GradScaler("cuda", enabled=use_amp)
        bce, l1 = torch.nn.BCEWithLogitsLoss(), torch.nn.L1Loss()
        lr = O.hash_input("amp.lr", (2, 3, 16, 16)).cuda()
        hr = O.hash_input("amp.hr", (2, 3, 64, 64)).cuda()
        with torch.autocast("cuda", dtype=torch.float16, enabled=use_amp):
            fake = g(lr)
            disc_real = d(hr)
            disc_fake = d(fake.detach())
            loss_disc = (bce(disc_real, torch.ones_like(disc_real)) +
                         bce(disc_fake, torch.zeros_like(disc_fake))) / 2
        d.zero_grad(set_to_none=True)
        scaler_d.scale(loss_disc).backward()
        scaler_d.step(opt_d)
        scaler_d.update()
        with torch.autocast("cuda", dtype=torch.float16, enabled=use_amp):
            disc_fake_for_gen = d(fake)
            loss_gen = l1(fake, hr) + 5e-3 * bce(disc_fake_for_gen, torch.ones_like(disc_fake_for_gen))
        g.zero_grad(set_to_none=True)
        scaler_g.scale(loss_gen).backward()
        g_grads = {n: p.grad.detach().clone() for n, p in g.named_parameters()}
        scaler_g.step(opt_g)
        scaler_g.update()
        torch.cuda.synchronize()
        return (float(loss_disc.detach()), float(loss_gen.detach()), g_grads,
                {n: p.detach().clone() for n, p in g.named_parameters()},
                {n: p.detach().clone() for n, p in d.named_parameters()},
                scaler_g.get_scale() if use_amp else 1.0)

    ld_a, lg_a, gg_a, pg_a, pd_a, scale = one_step(True)
    ld_b, lg_b, gg_b, pg_b, pd_b, _ = one_step(False)
    assert scale == 65536.0                           # no inf/nan was found: the scale did not back off
    assert abs(ld_a - ld_b) < 1e-5 and abs(lg_a - lg_b) < 1e-5
    for n in gg_a:
        assert torch.isfinite(gg_a[n]).all()
        # scaled gradients are the unscaled ones times 65536 (a power of two: exact in bf16 and
        # fp32), up to the atomics' summation order
        assert rel_l2(gg_a[n] / scale, gg_b[n]) < 1e-4, n
    for n in pg_a:
        assert rel_err(pg_a[n], pg_b[n]) < 1e-4, n
    for n in pd_a:
        assert rel_err(pd_a[n], pd_b[n]) < 1e-4, n
    # and the parameters did move
    g0 = _gen(blocks=1, salt=5)
    assert any(not torch.equal(p, pg_a[n]) for n, p in g0.named_parameters())


def test_half_model_eval_as_the_inference_engine_uses_it():
    """inference.py:116-118,153: model.eval(); model.half(); forward of a half input under
    autocast.  Output dtype follows the input; values are those of the bf16 compute path on the
    half-rounded parameters."""
    g = _gen(blocks=1, salt=9)
    x = O.hash_input("half.x", (1, 3, 20, 28)).cuda()
    gh = copy.deepcopy(g).eval().half()
    assert next(gh.parameters()).dtype == torch.float16
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        out = gh(x.half())
    assert out.dtype == torch.float16 and out.shape == (1, 3, 80, 112)
    assert torch.isfinite(out).all()
    # same parameters (rounded to half), explicit bf16 compute, fp32 interface
    g2 = copy.deepcopy(g)
    with torch.no_grad():
        for p in g2.parameters():
            p.copy_(p.half().float())
    g2.compute_dtype = torch.bfloat16
    with torch.no_grad():
        want = g2(x.half().float())
    assert torch.equal(out.float(), want.half().float())
    # state_dict of the half model keeps the reference's keys, in half
    assert list(gh.state_dict().keys()) == list(g.state_dict().keys())
    assert all(v.dtype == torch.float16 for v in gh.state_dict().values())


def test_discriminator_under_autocast_equals_bf16_compute():
    d1, d2 = _disc(4), _disc(4)
    x = O.hash_input("acd.x", (2, 3, 64, 64)).cuda()
    d2.compute_dtype = torch.bfloat16
    with torch.autocast("cuda", dtype=torch.float16):
        a = d1(x)
    b = d2(x)
    assert a.shape == (2, 1) and torch.equal(a.float(), b.float())
    # both advanced their power iteration once
    assert torch.equal(d1.state_dict()["features.0.0.weight_u"], d2.state_dict()["features.0.0.weight_u"])
