"""GPU parity of the two gradient kernels of a convolution, each as ONE call through the C ABI,
against torch.autograd of the oracle's convolution on the CPU, element by element over the whole
tensor: srx_conv2d_wgrad (+ bias gradient, + srx_unpack_wgrad) and the data-gradient form of
srx_conv2d (weights packed transposed and tap-flipped; stride 2 through srx_zero_stuff2).
Shapes are the layer shapes of esrgan/generator.py:10-14,43,54-62 and esrgan/discriminator.py:18-25."""
import pytest
import torch

from helpers import O, pkg, rel_err

pytestmark = pytest.mark.gpu

# per-tensor max-norm tolerance
GTOL = {torch.float32: 1e-3, torch.bfloat16: 2e-2}

CASES = [
    # B, cin, cout, H, W, stride
    (2, 64, 32, 20, 36, 1),
    (1, 96, 32, 17, 40, 1),
    (2, 128, 32, 12, 33, 1),
    (1, 160, 32, 33, 32, 1),
    (2, 192, 64, 16, 40, 1),
    (2, 3, 64, 19, 21, 1),
    (1, 64, 3, 40, 24, 1),
    (1, 64, 256, 10, 34, 1),
    (2, 64, 64, 32, 32, 2),
    (1, 64, 128, 16, 24, 2),
    (1, 3, 64, 14, 10, 1),
    (1, 64, 32, 136, 130, 1),
    # fewer rows than a tile has, a single row, one tile column exactly
    (1, 64, 32, 3, 24, 1),
    (3, 32, 32, 1, 57, 1),
    (1, 64, 64, 5, 64, 1),
]


def _reference(case, tdtype):
    B, cin, cout, H, W, stride = case
    x = O.hash_input(f"gx{case}", (B, cin, H, W)).requires_grad_(True)
    w = O.hash_uniform(f"gw{case}", (cout, cin, 3, 3), (3.0 / (9 * cin)) ** 0.5).requires_grad_(True)
    b = O.hash_uniform(f"gb{case}", (cout,), 0.1).requires_grad_(True)
    y = O.conv3x3(x, w, b, stride)
    gy = O.hash_input(f"ggy{case}", tuple(y.shape))
    gx, gw, gb = torch.autograd.grad(y, (x, w, b), gy)
    return x.detach(), w.detach(), gy, gx, gw, gb


@pytest.mark.parametrize("walk", [False, True])
@pytest.mark.parametrize("tdtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CASES)
def test_wgrad_full_tensor(case, tdtype, walk):
    """dW and db of nn.Conv2d: every element, against autograd; both kernels of the wide stride-1
    shapes (whole halo tiles, and the column walk that keeps input rows in an LDS ring)."""
    E = pkg("engine")
    B, cin, cout, H, W, stride = case
    if walk and (stride != 1 or W < 24):
        pytest.skip("the column-walk kernel serves stride-1 launches on images at least 24 pixels wide")
    x, w, gy, _, gw_ref, gb_ref = _reference(case, tdtype)
    dev = torch.device("cuda")
    plan = E.Plan(dev, tdtype)
    OH, OW = gy.shape[2], gy.shape[3]
    X, GY = plan.buf(B, cin, H, W), plan.buf(B, cout, OH, OW)
    plan.nchw_to_c8(x.to(dev), (X, 0))
    plan.nchw_to_c8(gy.to(dev), (GY, 0))
    stage_w, stage_b = plan.f32(cout * cin * 9), plan.f32(cout)
    dw, db = plan.f32(cout * cin * 9), plan.f32(cout)
    tasks = plan.wgrad_tasks_for_layer(cin, cout, 0, 0, stage_w.data_ptr(), bias_ptr=stage_b.data_ptr())
    ops = E.OpList()
    plan.wgrad_walk = walk        # walk=False: the plain kernel, whatever the engine's default
    plan.wgrad(ops, (X, 0), (GY, 0), tasks, B, H, W, ksize=3, stride=stride, tile_walk=walk)
    if walk:
        # several slices, so that slices begin in the middle of tile columns too
        ops[-1].u.wgrad.nslices = 7
    plan.unpack_item(stage_w.data_ptr(), dw.data_ptr(), cout, cin, 9)
    plan.unpack_item(stage_b.data_ptr(), db.data_ptr(), 1, cout, 1)
    plan.finalize_unpack()
    plan.run(plan.freeze(ops))
    plan.run_unpack()
    torch.cuda.synchronize()
    assert rel_err(dw.view(cout, cin, 3, 3), gw_ref) < GTOL[tdtype]
    assert rel_err(db, gb_ref) < GTOL[tdtype]


@pytest.mark.parametrize("tdtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CASES)
def test_dgrad_full_tensor(case, tdtype):
    """dX of nn.Conv2d as a convolution of dY with the transposed, tap-flipped weights."""
    E, L = pkg("engine"), pkg("_lib")
    B, cin, cout, H, W, stride = case
    if stride == 2 and (H % 2 or W % 2):
        pytest.skip("the stride-2 data gradient is built for even sizes")
    x, w, gy, gx_ref, _, _ = _reference(case, tdtype)
    dev = torch.device("cuda")
    plan = E.Plan(dev, tdtype)
    OH, OW = gy.shape[2], gy.shape[3]
    GY, GX = plan.buf(B, cout, OH, OW), plan.buf(B, cin, H, W)
    plan.nchw_to_c8(gy.to(dev), (GY, 0))
    wd = w.to(dev).contiguous()
    wp = plan.packed_layer(cin, cout, 9, 0, [E.dgrad_seg(wd, 0, 0)], with_bias=False)[0]
    plan.finalize_packing()
    plan.run_packing()
    ops = E.OpList()
    src = (GY, 0)
    if stride == 2:
        ZS = plan.buf(B, cout, H, W)
        plan.generic(ops, L.OP_ZERO_STUFF, a=(GY, 0), c=(ZS, 0), B=B, H=OH, W=OW, noct=(cout + 7) // 8)
        src = (ZS, 0)
    plan.conv(ops, src, (GX, 0), wp, 0, B, H, W, cout, cin)
    plan.run(plan.freeze(ops))
    gx = plan.c8_to_nchw((GX, 0), cin)
    torch.cuda.synchronize()
    assert gx.shape == gx_ref.shape
    assert rel_err(gx, gx_ref) < GTOL[tdtype]


def test_dense_block_stacked_dgrad_matches_autograd():
    """The data gradient of one growth slice of a dense block is ONE convolution over the stacked
    gradients [gy_{j+1} .. g_out] with the stacked transposed weights of convolutions j+1..5
    (several dgrad segments in one packed layer): against autograd of the oracle's block."""
    E = pkg("engine")
    dev = torch.device("cuda")
    B, H, W, nf, gc = 1, 12, 40, 64, 32
    x0 = O.hash_input("sd.x", (B, nf + gc, H, W))        # block input + first growth slice
    # convolutions 3, 4 (-> 32) and 5 (-> 64) all read slice x1 = channels [64, 96)
    ws = [O.hash_uniform(f"sd.w{k}", (gc if k < 5 else nf, nf + (k - 1) * gc, 3, 3), 0.05) for k in (3, 4, 5)]
    gys = [O.hash_input(f"sd.gy{k}", (B, gc if k < 5 else nf, H, W)) for k in (3, 4, 5)]
    # reference: sum over k of conv_transpose(gy_k, w_k)[channels 64..96)
    ref = torch.zeros(B, gc, H, W)
    for wk, gk in zip(ws, gys):
        xin = torch.zeros(B, wk.shape[1], H, W, requires_grad=True)
        (gxk,) = torch.autograd.grad(O.conv3x3(xin, wk, None), xin, gk)
        ref += gxk[:, nf:nf + gc]
    plan = E.Plan(dev, torch.float32)
    G = plan.buf(B, 2 * gc + nf, H, W)
    GX = plan.buf(B, gc, H, W)
    plan.nchw_to_c8(torch.cat(gys, 1).to(dev), (G, 0))
    wdev = [w.to(dev).contiguous() for w in ws]
    segs, c0 = [], 0
    for wk in wdev:
        segs.append(E.dgrad_seg(wk, c0, nf))
        c0 += wk.shape[0]
    wp = plan.packed_layer(gc, c0, 9, 0, segs, with_bias=False)[0]
    plan.finalize_packing()
    plan.run_packing()
    ops = E.OpList()
    plan.conv(ops, (G, 0), (GX, 0), wp, 0, B, H, W, c0, gc)
    plan.run(plan.freeze(ops))
    gx = plan.c8_to_nchw((GX, 0), gc)
    torch.cuda.synchronize()
    assert rel_err(gx, ref) < 1e-3


def test_column_walk_kernel_is_the_default_where_it_applies():
    """A plan left to itself takes the column-walk weight-gradient kernel for every 3x3 stride-1 launch
    on maps at least 24 pixels wide and the whole-tile kernel elsewhere; the descriptor says the same."""
    E, G, L = pkg("engine"), pkg("esrgan.generator"), pkg("_lib")
    from helpers import fill_module_
    assert E.Plan.wgrad_walk is True
    g = fill_module_(G.GeneratorESRGAN(num_blocks=1), salt=7).cuda()
    params = {n: p.data for n, p in g.named_parameters()}
    wide = E.GeneratorPlan(params, 2, 32, 32, torch.bfloat16, num_blocks=1, need_grad=True)
    narrow = E.GeneratorPlan(params, 2, 16, 16, torch.bfloat16, num_blocks=1, need_grad=True)
    for plan, trunk_w in ((wide, 32), (narrow, 16)):
        seen = set()
        for op, (name, _) in zip(plan.bwd_ops, plan.bwd_meta):
            if op.kind != L.OP_WGRAD:
                continue
            d = op.u.wgrad
            eligible = d.ksize == 3 and d.stride == 1 and d.W >= 24
            assert bool(d.tile_walk) == eligible
            assert name.startswith("wgrad_walk_kernel" if eligible else "wgrad_kernel")
            seen.add((d.W, bool(d.tile_walk)))
        assert (trunk_w, trunk_w >= 24) in seen          # the trunk's launches
        assert (4 * trunk_w, True) in seen               # the layers at HR resolution
