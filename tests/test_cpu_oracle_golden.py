"""CPU: the oracle restatement against the golden fixtures produced by the real reference
(tests/golden/make_golden.py).  Bar: exact equality where the reference used the same
ATen ops, 1e-6 relative where the restatement reorders arithmetic (spectral norm)."""
import numpy as np
import torch

from helpers import O, golden, rel_err

torch.set_num_threads(8)


def _sd(spec, salt):
    return O.make_state_dict(spec, salt=salt)


def test_pixel_shuffle_is_the_reference_permutation():
    _, arr = golden()
    x = torch.arange(2 * 8 * 3 * 5, dtype=torch.float32).reshape(2, 8, 3, 5)
    assert np.array_equal(O.pixel_shuffle2(x).numpy().astype(np.int32), arr["pixel_shuffle_out"])
    # index formula out[n,c,2y+i,2x+j] = in[n,4c+2i+j,y,x]
    y = O.pixel_shuffle2(x)
    for (n, c, yy, xx, i, j) in [(0, 0, 0, 0, 0, 0), (1, 1, 2, 4, 1, 0), (0, 1, 1, 3, 1, 1)]:
        assert y[n, c, 2 * yy + i, 2 * xx + j] == x[n, 4 * c + 2 * i + j, yy, xx]


def _block_spec(kind):
    def conv(name, cin, cout):
        return [(name + ".weight", (cout, cin, 3, 3)), (name + ".bias", (cout,))]
    rdb = lambda pre: sum([conv(pre + f"conv{k}", 64 + 32 * (k - 1), 32) for k in range(1, 5)], []) + \
        conv(pre + "conv5", 192, 64)
    if kind == "rdb":
        return rdb("")
    if kind == "rrdb":
        return rdb("rdb1.") + rdb("rdb2.") + rdb("rdb3.")
    return conv("conv", 64, 256)


def test_blocks_match_reference():
    man, arr = golden()
    for name, fn, shape in (("rdb", O.rdb_forward, (2, 64, 12, 20)), ("rrdb", O.rrdb_forward, (2, 64, 12, 20)),
                            ("upsample", O.upsample_forward, (2, 64, 6, 10))):
        sd = _sd(_block_spec(name), 1)
        x = O.hash_input(name + ".x", shape).requires_grad_(True)
        y = fn(x, sd)
        assert torch.equal(y.detach(), torch.from_numpy(arr[name + "_y"])), name
        gy = O.hash_input(name + ".gy", y.shape)
        (y * gy).sum().backward()
        assert rel_err(x.grad, torch.from_numpy(arr[name + "_gx"])) < 1e-5, name


def test_generator_one_block_matches_reference():
    man, arr = golden()
    case = man["cases"]["g1"]
    sd = _sd(O.generator_state_dict_spec(1), 2)
    for v in sd.values():
        v.requires_grad_(True)
    lr = O.hash_input("g1.lr", (2, 3, 16, 24))
    hr = O.hash_input("g1.hr", (2, 3, 64, 96))
    out = O.generator_forward(lr, sd)
    assert torch.equal(out.detach(), torch.from_numpy(arr["g1_out"]))
    loss = O.l1_loss(out, hr)
    assert abs(float(loss) - case["loss_value"]) < 1e-6
    grads = torch.autograd.grad(loss, list(sd.values()))
    for (k, _), g in zip(sd.items(), grads):
        ref = case["grads"][k]
        assert abs(float(g.double().norm()) - ref["norm"]) <= 1e-5 * ref["norm"] + 1e-12, k


def test_generator_23_blocks_config1_matches_reference():
    man, arr = golden()
    case = man["cases"]["g23"]
    sd = _sd(O.generator_state_dict_spec(23), 3)
    assert [[k, list(v.shape), str(v.dtype)] for k, v in sd.items()] == man["g_state_dict"]
    assert sum(v.numel() for v in sd.values()) == man["g_num_params"] == 16919555
    lr = O.hash_input("g23.lr", (4, 3, 32, 32))
    with torch.no_grad():
        out = O.generator_forward(lr, sd)
    assert torch.equal(out.flatten()[::7], torch.from_numpy(arr["g23_out_s7"]))
    assert abs(float(out.double().sum()) - case["out_sum"]) < 1e-6 * case["out_abs_sum"]


def test_discriminator_and_spectral_norm_match_reference():
    man, arr = golden()
    case = man["cases"]["d"]
    sd = _sd(O.discriminator_state_dict_spec(), 4)
    assert [[k, list(v.shape), str(v.dtype)] for k, v in sd.items()] == man["d_state_dict"]
    assert sum(v.numel() for k, v in sd.items() if k.endswith(("bias", "weight_orig"))) == man["d_num_params"]
    state = {k: v.clone() for k, v in sd.items() if k.endswith(("weight_u", "weight_v"))}
    for k in sd:
        if k.endswith(("bias", "weight_orig")):
            sd[k].requires_grad_(True)
    xr = O.hash_input("d.real", (2, 3, 64, 64))
    xf = O.hash_input("d.fake", (2, 3, 64, 64)).requires_grad_(True)
    lr_ = O.discriminator_forward(xr, sd, True, state)
    lf_ = O.discriminator_forward(xf, sd, True, state)
    assert rel_err(lr_, torch.from_numpy(arr["d_logits_real"])) < 1e-5
    assert rel_err(lf_, torch.from_numpy(arr["d_logits_fake"])) < 1e-5
    for k in ("features.0.0", "features.7.0", "classifier.1", "classifier.3"):
        for s in ("weight_u", "weight_v"):
            assert rel_err(state[f"{k}.{s}"], torch.from_numpy(arr[f"d_after2_{k}.{s}"])) < 1e-5
    loss = (O.bce_with_logits(lr_, 1.0) + O.bce_with_logits(lf_, 0.0)) / 2
    assert abs(float(loss) - case["loss_value"]) < 1e-6
    loss.backward()
    assert rel_err(xf.grad, torch.from_numpy(arr["d_gx_fake"])) < 1e-4
    for k, ref in case["grads"].items():
        assert abs(float(sd[k].grad.double().norm()) - ref["norm"]) <= 1e-4 * ref["norm"] + 1e-12, k
    l3 = O.discriminator_forward(xf.detach(), sd, True, state)
    assert rel_err(l3, torch.from_numpy(arr["d_logits_third"])) < 1e-5
    le = O.discriminator_forward(xr, sd, False, state)
    assert rel_err(le, torch.from_numpy(arr["d_logits_eval"])) < 1e-5


def test_discriminator_128_backward_matches_reference():
    """The discriminator at the reference's default HR crop (128 x 128, config.py:40-44): logits,
    BCE backward (whole gradient tensors, input gradient) and u / v after the pass."""
    man, arr = golden()
    case = man["cases"]["d128_bwd"]
    sd = _sd(O.discriminator_state_dict_spec(), 4)
    state = {k: v.clone() for k, v in sd.items() if k.endswith(("weight_u", "weight_v"))}
    for k in sd:
        if k.endswith(("bias", "weight_orig")):
            sd[k].requires_grad_(True)
    x = O.hash_input("d.x128", (2, 3, 128, 128)).requires_grad_(True)
    out = O.discriminator_forward(x, sd, True, state)
    assert rel_err(out, torch.from_numpy(arr["d_logits_128"])) < 1e-5
    loss = O.bce_with_logits(out, 1.0)
    assert abs(float(loss) - case["loss_value"]) < 1e-6
    loss.backward()
    assert rel_err(x.grad, torch.from_numpy(arr["d128_gx"])) < 1e-4
    for n, ent in case["full_grads"].items():
        ref = torch.from_numpy(arr[ent["array"]]).flatten()
        assert rel_err(sd[n].grad.flatten()[::ent["step"]], ref) < 1e-4, n
    for k in ("features.0.0", "features.6.0", "classifier.3"):
        for s_ in ("weight_u", "weight_v"):
            assert rel_err(state[f"{k}.{s_}"], torch.from_numpy(arr[f"d128_after1_{k}.{s_}"])) < 1e-5


def test_sn_backward_rank_one_identity():
    """dW_orig = dW_sn/sigma - (<dW_sn, W_orig>/sigma^2) u v^T  (what srx_sn_backward computes)."""
    w = O.hash_uniform("sn.w", (6, 4, 3, 3)).double().requires_grad_(True)
    u = torch.nn.functional.normalize(O.hash_uniform("sn.u", (6,)).double(), dim=0)
    v = torch.nn.functional.normalize(O.hash_uniform("sn.v", (36,)).double(), dim=0)
    wsn, u2, v2 = O.sn_weight(w, u, v, True)
    gw = O.hash_uniform("sn.g", (6, 4, 3, 3)).double()
    (wsn * gw).sum().backward()
    sigma = torch.dot(u2, w.detach().reshape(6, -1) @ v2)
    ident = gw / sigma - ((gw * w.detach()).sum() / sigma ** 2) * torch.outer(u2, v2).reshape(6, 4, 3, 3)
    assert float((ident - w.grad).abs().max()) < 1e-12


def test_adversarial_step_losses_match_reference():
    man, arr = golden()
    case = man["cases"]["adv_step"]
    g_sd = _sd(O.generator_state_dict_spec(2), 5)
    d_sd = _sd(O.discriminator_state_dict_spec(), 6)
    vgg_sd = O.make_state_dict(O.vgg19_state_dict_spec(), salt=7, gain=1.4)
    state = {k: v.clone() for k, v in d_sd.items() if k.endswith(("weight_u", "weight_v"))}
    lr = O.hash_input("adv.lr", (2, 3, 16, 16))
    hr = O.hash_input("adv.hr", (2, 3, 64, 64))
    with torch.no_grad():
        r = O.adversarial_step_losses(lr, hr, g_sd, d_sd, vgg_sd, state)
    assert torch.equal(r["fake"].flatten()[::5], torch.from_numpy(arr["adv_fake_s5"]))
    for k in ("loss_disc", "gen_adv_loss", "gen_l1_loss", "gen_vgg_loss", "loss_gen"):
        assert abs(float(r[k]) - case[k]) <= 1e-5 * abs(case[k]), k


def test_vgg_features_fixture():
    _, arr = golden()
    vgg_sd = O.make_state_dict(O.vgg19_state_dict_spec(), salt=7, gain=1.4)
    with torch.no_grad():
        f = O.vgg19_features36(O.hash_input("vgg.x", (1, 3, 32, 32)), vgg_sd)
    assert f.shape == (1, 512, 2, 2)
    assert rel_err(f, torch.from_numpy(arr["vgg_feat"])) < 1e-5
    assert sum(v.numel() for v in vgg_sd.values()) == 20024384
