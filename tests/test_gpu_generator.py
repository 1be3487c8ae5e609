"""GPU parity of the generator modules (drop-in nn.Module API) against the golden
fixtures produced by the real reference and against the CPU oracle."""
import pytest
import torch

from helpers import O, TOL, fill_module_, full_grads_bad, golden, pkg, rel_err, rel_l2

pytestmark = pytest.mark.gpu

# Whole gradient tensors, relative L2 against the reference's.  bf16: the derivative masks of
# LeakyReLU come from the bf16 forward, and on these few-hundred-pixel fixtures each sign flip of a
# near-zero activation moves a gradient tensor by O(1/pixels): 4-5e-2 observed.  The bf16 gradient
# kernels by themselves (masks taken from an fp32 forward) meet 2e-2 on every tensor of the
# 23-block generator: test_bf16_backward_with_fp32_masks_meets_tolerance_23_blocks.
FULL_GRAD_TOL = {torch.float32: 2e-3, torch.bfloat16: 6e-2}


def _grads_close(module, gold_grads, tol):
    bad = []
    for n, p in module.named_parameters():
        g = p.grad.detach().double().cpu().flatten()
        ref = gold_grads[n]
        nr = ref["norm"]
        err = abs(float(g.norm()) - nr) / max(nr, 1e-12)
        head = torch.tensor(ref["head"], dtype=torch.float64)
        herr = float((g[:8] - head).abs().max()) / max(nr / max(g.numel(), 1) ** 0.5, 1e-12)
        if err > tol or herr > 40 * tol:
            bad.append((n, err, herr))
    return bad


@pytest.mark.parametrize("tdtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("name", ["rdb", "rrdb", "upsample"])
def test_blocks_match_reference(name, tdtype):
    G = pkg("esrgan.generator")
    man, arr = golden()
    case = man["cases"][name]
    mod = {"rdb": lambda: G.ResidualDenseBlock(64, 32), "rrdb": lambda: G.RRDB(64),
           "upsample": lambda: G.UpsampleBlock(64)}[name]()
    fill_module_(mod, salt=1)
    mod = mod.cuda()
    mod.compute_dtype = tdtype
    shape = {"rdb": (2, 64, 12, 20), "rrdb": (2, 64, 12, 20), "upsample": (2, 64, 6, 10)}[name]
    x = O.hash_input(name + ".x", shape).cuda().requires_grad_(True)
    y = mod(x)
    ref_y = torch.from_numpy(arr[name + "_y"])
    assert rel_err(y, ref_y) < TOL[tdtype]
    gy = O.hash_input(name + ".gy", ref_y.shape).cuda()
    (y * gy).sum().backward()
    assert rel_err(x.grad, torch.from_numpy(arr[name + "_gx"])) < TOL[tdtype] * 2
    bad = _grads_close(mod, case["grads"], TOL[tdtype])
    assert not bad, bad[:5]
    bad = full_grads_bad(((n, p.grad) for n, p in mod.named_parameters()), case["full_grads"], arr, FULL_GRAD_TOL[tdtype])
    assert not bad, bad


@pytest.mark.parametrize("tdtype", [torch.float32, torch.bfloat16])
def test_generator_1block_matches_reference(tdtype):
    G = pkg("esrgan.generator")
    man, arr = golden()
    case = man["cases"]["g1"]
    g = fill_module_(G.GeneratorESRGAN(num_blocks=1), salt=2).cuda()
    g.compute_dtype = tdtype
    lr = O.hash_input("g1.lr", (2, 3, 16, 24)).cuda().requires_grad_(True)
    hr = O.hash_input("g1.hr", (2, 3, 64, 96)).cuda()
    out = g(lr)
    assert rel_err(out, torch.from_numpy(arr["g1_out"])) < TOL[tdtype]
    loss = torch.nn.L1Loss()(out, hr)
    assert abs(float(loss.detach()) - case["loss_value"]) / case["loss_value"] < TOL[tdtype]
    if tdtype == torch.float32:
        loss.backward()
    else:
        # sign(out-hr) is discontinuous: bf16 rounding flips it where out ~ hr, which is a
        # property of the L1 loss, not of the kernels.  Feed the reference's own dL/d(out).
        ref_out = torch.from_numpy(arr["g1_out"]).cuda()
        out.backward(torch.sign(ref_out - hr) / ref_out.numel())
    assert rel_l2(lr.grad, torch.from_numpy(arr["g1_glr"])) < TOL[tdtype] * 2
    bad = _grads_close(g, case["grads"], TOL[tdtype] * 2)
    assert not bad, bad[:5]
    bad = full_grads_bad(((n, p.grad) for n, p in g.named_parameters()), case["full_grads"], arr, FULL_GRAD_TOL[tdtype])
    assert not bad, bad


def test_generator_23_config1_fp32_matches_reference():
    """BASELINE config 1: 23 RRDB, batch 4 of 32x32 LR, fp32 — against the reference's output."""
    G = pkg("esrgan.generator")
    man, arr = golden()
    case = man["cases"]["g23"]
    g = fill_module_(G.GeneratorESRGAN(), salt=3).cuda()
    lr = O.hash_input("g23.lr", (4, 3, 32, 32)).cuda()
    hr = O.hash_input("g23.hr", (4, 3, 128, 128)).cuda()
    out = g(lr)
    samp = out.detach().flatten()[::7].cpu()
    assert rel_err(samp, torch.from_numpy(arr["g23_out_s7"])) < 1e-3
    assert abs(float(out.double().sum()) - case["out_sum"]) / case["out_abs_sum"] < 1e-4
    loss = torch.nn.L1Loss()(out, hr)
    assert abs(float(loss) - case["loss_value"]) / case["loss_value"] < 1e-3
    loss.backward()
    bad = _grads_close(g, case["grads"], 2e-3)
    assert not bad, bad[:5]
    bad = full_grads_bad(((n, p.grad) for n, p in g.named_parameters()), case["full_grads"], arr, 2e-3)
    assert not bad, bad


def test_generator_23_bf16_forward_close():
    G = pkg("esrgan.generator")
    man, arr = golden()
    g = fill_module_(G.GeneratorESRGAN(), salt=3).cuda()
    g.compute_dtype = torch.bfloat16
    lr = O.hash_input("g23.lr", (4, 3, 32, 32)).cuda()
    with torch.no_grad():
        out = g(lr)
    samp = out.flatten()[::7].cpu()
    ref = torch.from_numpy(arr["g23_out_s7"])
    assert rel_l2(samp, ref) < 2e-2


def test_config2_full_size_homogeneity_and_determinism():
    """Size-independent properties at the benchmark's full size (B=16, 64x64 LR, 23 blocks, bf16):
    with zero biases the generator is positively homogeneous (LeakyReLU and the residual
    scalings commute with a power-of-two scale, and so does bf16 rounding), so G(2x) == 2 G(x)
    bit for bit; and the forward is deterministic."""
    E = pkg("engine")
    G = pkg("esrgan.generator")
    g = fill_module_(G.GeneratorESRGAN(num_blocks=23), salt=13).cuda()
    with torch.no_grad():
        for n, p in g.named_parameters():
            if n.endswith(".bias"):
                p.zero_()
    params = {n: p.data for n, p in g.named_parameters()}
    plan = E.GeneratorPlan(params, 16, 64, 64, torch.bfloat16, num_blocks=23, need_grad=False)
    x = (O.hash_input("c2.x", (16, 3, 64, 64)) * 0.5).cuda()
    y1 = plan.forward(x).clone()
    y1b = plan.forward(x).clone()
    y2 = plan.forward(2 * x).clone()
    assert y1.shape == (16, 3, 256, 256)
    assert torch.isfinite(y1).all() and float(y1.abs().max()) > 0
    assert torch.equal(y1, y1b)
    assert torch.equal(y2, 2 * y1)


def test_paired_weight_gradient_tasks_give_the_same_gradients():
    """Tasks that pair two distant 32-channel groups (x_skip / gy_skip of srx_wgrad_task) compute the
    same weight gradients as the plain task list."""
    E = pkg("engine")
    G = pkg("esrgan.generator")
    x = O.hash_input("pw.x", (2, 3, 16, 24)).cuda()
    hr = O.hash_input("pw.hr", (2, 3, 64, 96)).cuda()
    grads = []
    for flag in (False, True):
        E.Plan.pair_wgrad_tasks = flag
        try:
            g = fill_module_(G.GeneratorESRGAN(num_blocks=2), salt=21).cuda()
            loss = torch.nn.functional.l1_loss(g(x), hr)
            loss.backward()
            grads.append({n: p.grad.clone() for n, p in g.named_parameters()})
        finally:
            E.Plan.pair_wgrad_tasks = False
    for n in grads[0]:
        assert rel_err(grads[1][n], grads[0][n]) < 1e-5, n


def _oracle_fwd_bwd(g, lr, gy):
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in g.state_dict().items()}
    ref = O.generator_forward(lr, sd)
    ref.backward(gy)
    return ref.detach(), {k: v.grad for k, v in sd.items()}


@pytest.mark.parametrize("tdtype", [torch.float32, torch.bfloat16])
def test_config5_shape_one_block_matches_oracle(tdtype):
    """128x128 LR patches (BASELINE config 5's patch size; every dense-block convolution then runs
    on the 32x16-pixel tile shape): one RRDB, batch 2, forward and every parameter's whole gradient
    tensor against the oracle on the CPU."""
    G = pkg("esrgan.generator")
    E = pkg("engine")
    g = fill_module_(G.GeneratorESRGAN(num_blocks=1), salt=17).cuda()
    lr = O.hash_input("c5.lr", (2, 3, 128, 128))
    gy = O.hash_input("c5.gy", (2, 3, 512, 512)) / (2 * 3 * 512 * 512)
    ref, ref_grads = _oracle_fwd_bwd(g, lr, gy)
    params = {n: p.data for n, p in g.named_parameters()}
    # (layer by layer: a batch of 2 is small enough for the one-launch dense-block forward, which
    #  tests/test_gpu_rdb_fused.py covers; this test is about the tile shape of the large batches)
    keep, E._GenBase.fused_rdb = E._GenBase.fused_rdb, False
    try:
        plan = E.GeneratorPlan(params, 2, 128, 128, tdtype, num_blocks=1, need_grad=True)
    finally:
        E._GenBase.fused_rdb = keep
    dt = 0 if tdtype == torch.bfloat16 else 1
    # the 12 growth convolutions of the three dense blocks (and conv_last, 64 -> 3 at 512 x 512)
    assert sum(n == E.conv_kernel_name(dt, 3, 1, 128, 128, 32, 2) and n.endswith(",1,1,4,5,3>")
               for n, _ in plan.fwd_meta) == 13
    out = plan.forward(lr.cuda())
    plan.backward(gy.cuda())
    torch.cuda.synchronize()
    assert rel_err(out, ref) < TOL[tdtype]
    if tdtype == torch.float32:
        bad = {n: e for n in plan.param_names if not (e := rel_l2(plan.grads[n], ref_grads[n])) < 2e-3}
        assert not bad, bad
        return
    # bf16, masks from its own forward: about 1 % of the near-zero activations change sign under
    # bf16 rounding, and with this zero-mean output gradient each flip is an incoherent O(1)
    # change of one term: 6-9e-2 relative L2 per tensor, the direction is kept
    a = torch.cat([plan.grads[n].flatten() for n in plan.param_names]).double().cpu()
    b = torch.cat([ref_grads[n].flatten() for n in plan.param_names]).double()
    assert float(torch.dot(a, b) / (a.norm() * b.norm())) > 0.99
    # the bf16 kernels themselves: with the saved activations of an fp32 forward in the plan's
    # buffers (same signs, hence the oracle's masks) every tensor meets the bf16 tolerance
    p32k = E.GeneratorPlan(params, 2, 128, 128, torch.float32, num_blocks=1, need_grad=True)
    p32k.forward(lr.cuda())
    for name in ("X0", "T1", "U0", "U1", "V"):
        getattr(plan, name).t.copy_(getattr(p32k, name).t)
    for r16, r32 in zip(plan.R, p32k.R):
        r16.t.copy_(r32.t)
    plan.backward(gy.cuda())
    torch.cuda.synchronize()
    bad = {n: e for n in plan.param_names if not (e := rel_l2(plan.grads[n], ref_grads[n])) < 2e-2}
    assert not bad, bad


def test_config5_full_size_forward_properties():
    """BASELINE config 5 at its full size (B=64 of 128x128 LR, 23 blocks, bf16), forward:
    finite, deterministic, and — with zero biases — positively homogeneous bit for bit
    (G(2x) == 2 G(x): LeakyReLU, the residual scalings and bf16 rounding commute with a power of
    two), and batch-shard independent: image i of the batch equals the same image run alone in a
    batch of 2 (the property batch sharding over GPUs relies on)."""
    E = pkg("engine")
    G = pkg("esrgan.generator")
    g = fill_module_(G.GeneratorESRGAN(num_blocks=23), salt=13).cuda()
    with torch.no_grad():
        for n, p in g.named_parameters():
            if n.endswith(".bias"):
                p.zero_()
    params = {n: p.data for n, p in g.named_parameters()}
    plan = E.GeneratorPlan(params, 64, 128, 128, torch.bfloat16, num_blocks=23, need_grad=False)
    x = (O.hash_input("c5f.x", (64, 3, 128, 128)) * 0.5).cuda()
    y1 = plan.forward(x).clone()
    y1b = plan.forward(x)
    assert y1.shape == (64, 3, 512, 512)
    assert torch.isfinite(y1).all() and float(y1.abs().max()) > 0
    assert torch.equal(y1, y1b)
    del y1b
    y2 = plan.forward(2 * x)
    assert torch.equal(y2, 2 * y1)
    del y2, plan
    small = E.GeneratorPlan(params, 2, 128, 128, torch.bfloat16, num_blocks=23, need_grad=False)
    ys = small.forward(x[[5, 63]].contiguous())
    assert torch.equal(ys, y1[[5, 63]])


def test_bf16_backward_with_fp32_masks_meets_tolerance_23_blocks():
    """The bf16 backward kernels in isolation, 23 blocks, BASELINE config 1's shape: the saved
    activations of an fp32 forward (rounded to bf16, signs unchanged) are placed in the bf16 plan's
    buffers, so its activation-derivative masks are those of the fp32 run; every parameter's whole
    gradient tensor must then agree with the fp32 plan's (itself pinned against the reference's
    gradients) within the bf16 tolerance.  The ~10 % deviation of bf16 training gradients on this
    network is therefore a property of masks taken from a bf16 forward, not of the kernels."""
    E = pkg("engine")
    G = pkg("esrgan.generator")
    g = fill_module_(G.GeneratorESRGAN(num_blocks=23), salt=3).cuda()
    params = {n: p.data for n, p in g.named_parameters()}
    lr = O.hash_input("g23.lr", (4, 3, 32, 32)).cuda()
    hr = O.hash_input("g23.hr", (4, 3, 128, 128)).cuda()
    p32 = E.GeneratorPlan(params, 4, 32, 32, torch.float32, num_blocks=23, need_grad=True)
    p16 = E.GeneratorPlan(params, 4, 32, 32, torch.bfloat16, num_blocks=23, need_grad=True)
    out = p32.forward(lr)
    gy = torch.sign(out - hr) / out.numel()
    p32.backward(gy)
    p16.forward(lr)                      # packs the weights; its activations are replaced below
    saved = ["X0", "T1", "U0", "U1", "V"]
    for name in saved:
        getattr(p16, name).t.copy_(getattr(p32, name).t)
    for a, b in zip(p16.R, p32.R):
        a.t.copy_(b.t)
    p16.backward(gy)
    torch.cuda.synchronize()
    bad = {}
    for n in p32.param_names:
        e = rel_l2(p16.grads[n], p32.grads[n])
        if not e < 2e-2:
            bad[n] = e
    assert not bad, (len(bad), sorted(bad.items(), key=lambda kv: -kv[1])[:8])


def test_config2_full_size_backward_directional_derivative():
    """Backward at the benchmark's full size (B=16 of 64x64 LR, 23 blocks), fp32 MFMA: for the
    linear loss L = <out, gy> and a random parameter direction d, the central finite difference
    (L(theta + eps d) - L(theta - eps d)) / (2 eps) equals <grad, d>.  A size-independent check that
    ties the gradient kernels to the forward kernels at the real shape; then the bf16 gradient of
    the same step must point the same way as the fp32 one."""
    E = pkg("engine")
    G = pkg("esrgan.generator")
    g = fill_module_(G.GeneratorESRGAN(num_blocks=23), salt=29).cuda()
    params = {n: p.data for n, p in g.named_parameters()}
    x = O.hash_input("fd.x", (16, 3, 64, 64)).cuda()
    gy = O.hash_input("fd.gy", (16, 3, 256, 256)).cuda() / (16 * 3 * 256 * 256)
    plan = E.GeneratorPlan(params, 16, 64, 64, torch.float32, num_blocks=23, need_grad=True)
    plan.forward(x)
    plan.backward(gy)
    grads = {n: plan.grads[n].clone() for n in plan.param_names}
    gen = torch.Generator(device="cpu").manual_seed(5)
    dirs = {n: (torch.randn(p.shape, generator=gen) * p.detach().abs().mean().cpu()).cuda() for n, p in params.items()}
    analytic = sum(float((grads[n].double() * dirs[n].double()).sum()) for n in grads)
    eps = 1e-2

    def loss_at(sign):
        for n, p in params.items():
            p.add_(dirs[n], alpha=sign * eps)
        out = plan.forward(x)
        val = float((out.double() * gy.double()).sum())
        for n, p in params.items():
            p.add_(dirs[n], alpha=-sign * eps)
        return val
    fd = (loss_at(+1) - loss_at(-1)) / (2 * eps)
    assert abs(fd - analytic) / max(abs(analytic), 1e-30) < 2e-2, (fd, analytic)
    del plan
    p16 = E.GeneratorPlan(params, 16, 64, 64, torch.bfloat16, num_blocks=23, need_grad=True)
    p16.forward(x)
    p16.backward(gy)
    torch.cuda.synchronize()
    a = torch.cat([p16.grads[n].flatten() for n in p16.param_names]).double()
    b = torch.cat([grads[n].flatten() for n in p16.param_names]).double()
    assert torch.isfinite(a).all()
    assert float(torch.dot(a, b) / (a.norm() * b.norm())) > 0.97


def test_deterministic_weight_gradients_same_bits_twice():
    """Plan.deterministic_wgrad: every pixel slice stores its partial sums into its own copy of the
    staging buffer and the unpack launch adds the copies in slice order, so two runs on the same
    inputs give the same bits for every gradient (bf16 and fp32, the two-stream backward included),
    and the values are those of the default atomic-add form up to summation order."""
    E = pkg("engine")
    G = pkg("esrgan.generator")
    g = fill_module_(G.GeneratorESRGAN(num_blocks=3), salt=23).cuda()
    params = {n: p.data for n, p in g.named_parameters()}
    x = O.hash_input("det.x", (4, 3, 40, 48)).cuda()
    gy = O.hash_input("det.gy", (4, 3, 160, 192)).cuda() / (4 * 3 * 160 * 192)
    for tdtype in (torch.bfloat16, torch.float32):
        ref_plan = E.GeneratorPlan(params, 4, 40, 48, tdtype, num_blocks=3, need_grad=True)
        ref_plan.forward(x)
        ref_plan.backward(gy)
        ref = ref_plan.flat_grad.clone()
        E.Plan.deterministic_wgrad = True
        try:
            plan = E.GeneratorPlan(params, 4, 40, 48, tdtype, num_blocks=3, need_grad=True)
        finally:
            E.Plan.deterministic_wgrad = False
        assert plan.stage.numel() == plan.det_slices * ref_plan.stage.numel()
        assert max(it.nslices for it in plan.unpack_items) > 1
        runs = []
        for _ in range(3):
            plan.forward(x)
            plan.backward(gy)
            torch.cuda.synchronize()
            runs.append(plan.flat_grad.clone())
        assert torch.equal(runs[0], runs[1]) and torch.equal(runs[1], runs[2])
        assert float(runs[0].abs().max()) > 0
        assert rel_err(runs[0], ref) < 1e-5


def test_deterministic_weight_gradients_discriminator():
    E = pkg("engine")
    CH = pkg("chain")
    D = pkg("esrgan.discriminator")
    d = fill_module_(D.Discriminator(), salt=6).cuda()
    params = {n: p.data for n, p in d.named_parameters()}
    bufs = {n: b.clone() for n, b in d.named_buffers()}
    x = O.hash_input("detd.x", (4, 3, 64, 64)).cuda()
    gl = torch.tensor([0.3, -0.2, 0.1, 0.25], device="cuda")
    E.Plan.deterministic_wgrad = True
    try:
        plan = CH.DiscriminatorPlan(params, bufs, 4, 64, 64, torch.bfloat16, need_param_grad=True, need_input_grad=False)
    finally:
        E.Plan.deterministic_wgrad = False
    runs = []
    saved = {n: b.clone() for n, b in bufs.items()}
    for _ in range(2):
        for n, b in bufs.items():          # the power iteration advances u, v in place: same start for both runs
            b.copy_(saved[n])
        plan.forward_logits(x, training=True)
        plan.backward(gl)
        torch.cuda.synchronize()
        runs.append(plan.flat_grad.clone())
    assert torch.equal(runs[0], runs[1]) and float(runs[0].abs().max()) > 0
