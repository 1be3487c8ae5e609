"""The fused dense-block forward (srx_rdb_forward, one launch per ResidualDenseBlock) and the fused
data gradient of the block (srx_rdb_backward) against the layer-by-layer path (five srx_conv2d
launches each): the same bits, for training plans (growth slices stored for the backward pass) and
inference plans (only the block output written), on image sizes that are and are not multiples of
the 16 x 8 tile, including images smaller than a tile."""
import pytest
import torch

from helpers import O, fill_module_, pkg

pytestmark = pytest.mark.gpu


def _plans(shape, need_grad, num_blocks=2, fused_bwd=False, tile=0):
    """(layer-by-layer plan, plan with the fused forward [and the fused data gradient]) on 16 x 8
    (tile 0), 8 x 8 (tile 1) or two stacked 16 x 8 (tile 2) pixel tiles"""
    E, G = pkg("engine"), pkg("esrgan.generator")
    g = fill_module_(G.GeneratorESRGAN(num_blocks=num_blocks), salt=51).cuda()
    params = {n: p.data for n, p in g.named_parameters()}
    B, h, w = shape
    out, keep = [], (E._GenBase.fused_rdb, E._GenBase.fused_rdb_bwd, E._GenBase.rdb_tile_fwd, E._GenBase.rdb_tile_bwd)
    for fused in (False, True):
        E._GenBase.fused_rdb, E._GenBase.fused_rdb_bwd = fused, fused and fused_bwd
        E._GenBase.rdb_tile_fwd = E._GenBase.rdb_tile_bwd = tile
        try:
            out.append(E.GeneratorPlan(params, B, h, w, torch.bfloat16, num_blocks=num_blocks, need_grad=need_grad))
        finally:
            (E._GenBase.fused_rdb, E._GenBase.fused_rdb_bwd, E._GenBase.rdb_tile_fwd, E._GenBase.rdb_tile_bwd) = keep
    names = [[m[0] for m in p.fwd_meta] for p in out]
    assert "rdb_fwd_kernel" not in names[0] and names[1].count("rdb_fwd_kernel") == 3 * num_blocks
    if need_grad:
        names = [[m[0] for m in p.bwd_meta] for p in out]
        assert "rdb_bwd_kernel" not in names[0]
        assert names[1].count("rdb_bwd_kernel") == (3 * num_blocks if fused_bwd else 0)
    return out


@pytest.mark.parametrize("tile", [0, 1, 2])
@pytest.mark.parametrize("need_grad", [True, False])
@pytest.mark.parametrize("shape", [(2, 20, 28), (1, 37, 53), (3, 64, 64), (2, 5, 9), (1, 8, 16), (2, 16, 8)])
def test_fused_block_forward_equals_layer_by_layer(shape, need_grad, tile):
    ref, fus = _plans(shape, need_grad, tile=tile)
    x = O.hash_input("fr.x", (shape[0], 3, shape[1], shape[2])).cuda()
    y0 = ref.forward(x)
    y1 = fus.forward(x)
    torch.cuda.synchronize()
    assert torch.isfinite(y0).all() and float(y0.abs().max()) > 0
    assert torch.equal(y0, y1)
    if need_grad:
        # every dense block's buffer: input planes and the four growth slices the backward pass reads
        # (the last buffer only ever receives the trunk's output in its first 8 planes)
        B, hw8 = shape[0], shape[1] * shape[2] * 8
        for r, (a, b) in enumerate(zip(ref.R, fus.R)):
            planes = 24 if r + 1 < len(ref.R) else 8
            va = a.t.view(torch.int16).view(B, 24, hw8)[:, :planes]
            vb = b.t.view(torch.int16).view(B, 24, hw8)[:, :planes]
            assert torch.equal(va, vb), f"block buffer {r} differs"


def test_fused_block_backward_uses_the_stored_growth_slices():
    """Forward fused, backward layer by layer: gradients equal those of the all-layer-by-layer plan
    up to the summation order of the weight-gradient atomics."""
    ref, fus = _plans((2, 24, 40), True)
    x = O.hash_input("fr.x2", (2, 3, 24, 40)).cuda()
    gy = O.hash_input("fr.gy", (2, 3, 96, 160)).cuda() / (2 * 3 * 96 * 160)
    for p in (ref, fus):
        p.forward(x)
        p.backward(gy)
    torch.cuda.synchronize()
    a, b = ref.flat_grad, fus.flat_grad
    assert float(a.abs().max()) > 0
    assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max())


@pytest.mark.parametrize("tile", [0, 1, 2])
@pytest.mark.parametrize("shape", [(2, 20, 28), (1, 37, 53), (3, 64, 64), (2, 5, 9), (1, 8, 16), (2, 16, 8)])
def test_fused_block_data_gradient_equals_layer_by_layer(shape, tile):
    """srx_rdb_backward against the five data-gradient launches: every gradient buffer
    (gy1..gy4 | g_out of every block) and the gradient reaching conv_first's output bit for bit;
    the parameter gradients up to the summation order of the weight-gradient atomics."""
    ref, fus = _plans(shape, True, fused_bwd=True, tile=tile)
    B, h, w = shape
    x = O.hash_input("fb.x", (B, 3, h, w)).cuda()
    gy = O.hash_input("fb.gy", (B, 3, 4 * h, 4 * w)).cuda() / (B * 3 * 16 * h * w)
    for p in (ref, fus):
        p.forward(x)
        p.backward(gy)
    torch.cuda.synchronize()
    for r, (a, b) in enumerate(zip(ref.G, fus.G)):
        va, vb = a.t.view(torch.int16), b.t.view(torch.int16)
        assert float(a.t.float().abs().max()) > 0
        assert torch.equal(va, vb), f"gradient buffer {r} differs"
    assert torch.equal(ref.gfeat.t.view(torch.int16), fus.gfeat.t.view(torch.int16))
    a, b = ref.flat_grad, fus.flat_grad
    assert float(a.abs().max()) > 0
    assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max())


def test_fused_block_rejects_what_it_does_not_cover():
    L = pkg("_lib")
    lib = L.lib()
    d = L.RdbDesc(dtype=L.F32, B=1, H=8, W=8, nf=64, gc=32)
    assert lib.srx_rdb_forward(d, None) == 1 and b"bf16 only" in lib.srx_last_error()
    d = L.RdbDesc(dtype=L.BF16, B=1, H=8, W=8, nf=32, gc=16)
    assert lib.srx_rdb_forward(d, None) == 1 and b"64 block channels" in lib.srx_last_error()
    assert lib.srx_rdb_forward(None, None) == 1
    d = L.RdbDesc(dtype=L.BF16, B=1, H=8, W=8, nf=64, gc=32)
    assert lib.srx_rdb_backward(d, None) == 1 and b"null buf" in lib.srx_last_error()
    assert lib.srx_rdb_backward(None, None) == 1
    d = L.RdbDesc(dtype=L.BF16, B=1, H=8, W=8, nf=64, gc=32, tile=3, buf=256, y=256, w=256, bias=256)
    assert lib.srx_rdb_forward(d, None) == 1 and b"tile must be" in lib.srx_last_error()


def test_fused_block_is_chosen_by_size():
    """Left to itself a plan takes the fused forward while its tiles fit the chip in two rounds, the
    fused data gradient while they fit in one."""
    E, G = pkg("engine"), pkg("esrgan.generator")
    g = fill_module_(G.GeneratorESRGAN(num_blocks=1), salt=52).cuda()
    params = {n: p.data for n, p in g.named_parameters()}
    assert E._GenBase.fused_rdb is None
    small = E.GeneratorPlan(params, 16, 32, 32, torch.bfloat16, num_blocks=1, need_grad=False)
    large = E.GeneratorPlan(params, 32, 64, 64, torch.bfloat16, num_blocks=1, need_grad=False)
    f32 = E.GeneratorPlan(params, 2, 16, 16, torch.float32, num_blocks=1, need_grad=False)
    assert [m[0] for m in small.fwd_meta].count("rdb_fwd_kernel") == 3
    assert all(m[0] != "rdb_fwd_kernel" for m in large.fwd_meta + f32.fwd_meta)
    assert E._GenBase.fused_rdb_bwd is None
    t128 = E.GeneratorPlan(params, 16, 32, 32, torch.bfloat16, num_blocks=1, need_grad=True)
    t512 = E.GeneratorPlan(params, 16, 64, 64, torch.bfloat16, num_blocks=1, need_grad=True)
    assert [m[0] for m in t128.bwd_meta].count("rdb_bwd_kernel") == 3
    assert [m[0] for m in t512.fwd_meta].count("rdb_fwd_kernel") == 3
    assert all(m[0] != "rdb_bwd_kernel" for m in t512.bwd_meta)
    # tiles: training plans 16 x 8, stacked in pairs where there are more tiles than CUs; inference
    # plans of small maps 8 x 8
    L = pkg("_lib")
    tile = lambda plan, ops, kind: [op.u.rdb.tile for op in ops if op.kind == kind]
    assert set(tile(t128, t128.fwd_ops, L.OP_RDB_FWD)) == {0} and set(tile(t128, t128.bwd_ops, L.OP_RDB_BWD)) == {0}
    assert set(tile(t512, t512.fwd_ops, L.OP_RDB_FWD)) == {2}
    assert set(tile(small, small.fwd_ops, L.OP_RDB_FWD)) == {1}
