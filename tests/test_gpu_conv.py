"""GPU parity of the HIP convolution (C ABI srx_conv2d) against the CPU oracle."""
import os

import pytest
import torch

from helpers import O, TOL, pkg, rel_err, run_single_conv

pytestmark = pytest.mark.gpu

CASES = [
    # B, cin, cout, H, W, stride
    (2, 64, 32, 12, 20, 1),
    (1, 96, 32, 33, 40, 1),
    (2, 192, 64, 16, 64, 1),
    (1, 3, 64, 17, 9, 1),
    (1, 64, 3, 40, 24, 1),
    (2, 64, 64, 32, 32, 2),
    (1, 128, 128, 16, 16, 2),
    (1, 256, 256, 8, 8, 1),
    (1, 3, 64, 64, 64, 2),
    (1, 16, 48, 8, 8, 1),
    (1, 64, 32, 130, 70, 1),
]


@pytest.mark.parametrize("tdtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CASES)
def test_conv3x3_bias_lrelu(case, tdtype):
    B, cin, cout, H, W, stride = case
    x = O.hash_input(f"x{case}", (B, cin, H, W))
    w = O.hash_uniform(f"w{case}", (cout, cin, 3, 3), (3.0 / (9 * cin)) ** 0.5)
    b = O.hash_uniform(f"b{case}", (cout,), 0.1)
    ref = O.lrelu(O.conv3x3(x, w, b, stride))
    out = run_single_conv(x, w, b, stride=stride, tdtype=tdtype, act=0.2)
    assert out.shape == ref.shape
    assert rel_err(out, ref) < TOL[tdtype]


# Dense-block shapes on images of at least 128 x 128 output pixels: the four-row-group tile shape
# (32 x 16 output pixels per workgroup) that large patches run on (esrgan/generator.py:10-13 at
# 128 x 128 LR patches), ragged sizes included.
BIG_CASES = [
    # B, cin, cout, H, W
    (1, 64, 32, 128, 128),
    (1, 96, 32, 128, 128),
    (1, 128, 32, 144, 128),
    (1, 160, 32, 128, 160),
    (2, 64, 32, 130, 136),
]


@pytest.mark.parametrize("tdtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("epilogue", ["bias_lrelu", "mask"])
@pytest.mark.parametrize("case", BIG_CASES)
def test_conv3x3_large_image_dense_shapes(case, epilogue, tdtype):
    B, cin, cout, H, W = case
    E = pkg("engine")
    assert E.conv_kernel_name(0 if tdtype == torch.bfloat16 else 1, 3, 1, H, W, cout, B).endswith(",1,1,4,5,3>")
    x = O.hash_input(f"bx{case}", (B, cin, H, W))
    w = O.hash_uniform(f"bw{case}", (cout, cin, 3, 3), (3.0 / (9 * cin)) ** 0.5)
    if epilogue == "bias_lrelu":
        b = O.hash_uniform(f"bb{case}", (cout,), 0.1)
        ref = O.lrelu(O.conv3x3(x, w, b))
        out = run_single_conv(x, w, b, tdtype=tdtype, act=0.2)
    else:
        # data-gradient form: no bias, output multiplied by the activation derivative of a saved tensor
        mask = O.hash_input(f"bm{case}", (B, cout, H, W))
        ref = O.conv3x3(x, w, None) * torch.where(mask > 0, 1.0, 0.2)
        out = run_single_conv(x, w, None, tdtype=tdtype, mask=mask, mask_slope=0.2)
    assert out.shape == ref.shape
    assert rel_err(out, ref) < TOL[tdtype]


# Grids that take the XCD-aware workgroup -> tile map of the launcher: 64 or more workgroups with
# power-of-two grid extents (shift decode; the second case has two output-block groups in grid x),
# and 1 024 or more with ragged extents (division decode, 1 053 = 8 * 131 + 5: the last five
# workgroups keep their launch-order tile).
XCD_MAP_CASES = [
    (4, 16, 32, 64, 64),
    (2, 16, 128, 64, 64),
    (9, 8, 32, 200, 288),
]


@pytest.mark.parametrize("tdtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", XCD_MAP_CASES)
def test_conv3x3_xcd_tile_map(case, tdtype):
    B, cin, cout, H, W = case
    x = O.hash_input(f"mx{case}", (B, cin, H, W))
    w = O.hash_uniform(f"mw{case}", (cout, cin, 3, 3), (3.0 / (9 * cin)) ** 0.5)
    b = O.hash_uniform(f"mb{case}", (cout,), 0.1)
    ref = O.lrelu(O.conv3x3(x, w, b))
    out = run_single_conv(x, w, b, tdtype=tdtype, act=0.2)
    assert out.shape == ref.shape
    assert rel_err(out, ref) < TOL[tdtype]
    # every image and tile separately: a wrong map would swap whole tiles between images
    for bi in range(B):
        assert rel_err(out[bi], ref[bi]) < 2 * TOL[tdtype]


@pytest.mark.parametrize("tdtype", [torch.float32, torch.bfloat16])
def test_conv_residual_mask(tdtype):
    x = O.hash_input("xr", (2, 192, 10, 36))
    w = O.hash_uniform("wr", (64, 192, 3, 3), (3.0 / (9 * 192)) ** 0.5)
    b = O.hash_uniform("br", (64,), 0.1)
    res = O.hash_input("res", (2, 64, 10, 36))
    mask = O.hash_input("mask", (2, 64, 10, 36))
    y = O.conv3x3(x, w, b)
    ref = 0.2 * (y * torch.where(mask > 0, 1.0, 0.3)) + 0.5 * res
    out = run_single_conv(x, w, b, tdtype=tdtype, alpha=0.2, res=res, beta=0.5, mask=mask, mask_slope=0.3)
    assert rel_err(out, ref) < TOL[tdtype]


@pytest.mark.parametrize("tdtype", [torch.float32, torch.bfloat16])
def test_conv_pixel_shuffle(tdtype):
    x = O.hash_input("xp", (2, 64, 9, 33))
    w = O.hash_uniform("wp", (256, 64, 3, 3), (3.0 / (9 * 64)) ** 0.5)
    b = O.hash_uniform("bp", (256,), 0.1)
    ref = O.lrelu(O.pixel_shuffle2(O.conv3x3(x, w, b)))
    out = run_single_conv(x, w, b, tdtype=tdtype, act=0.2, ps=1)
    assert out.shape == ref.shape
    assert rel_err(out, ref) < TOL[tdtype]


def test_pixel_shuffle_indexing_bit_exact():
    """Integer-valued data through an identity centre-tap conv: the PixelShuffle scatter
    must reproduce nn.PixelShuffle(2) exactly (reference esrgan/generator.py:44)."""
    import numpy as np
    from helpers import golden
    man, arr = golden()
    cin = cout = 32
    x = (torch.arange(1 * cin * 5 * 7) % 251).float().reshape(1, cin, 5, 7)
    w = torch.zeros(cout, cin, 3, 3)
    for c in range(cout):
        w[c, c, 1, 1] = 1.0
    for tdtype in (torch.float32, torch.bfloat16):
        out = run_single_conv(x, w, None, tdtype=tdtype, ps=1)
        assert torch.equal(out, O.pixel_shuffle2(x))
    # and the oracle's index formula equals the reference's nn.PixelShuffle output (golden)
    xg = torch.arange(2 * 8 * 3 * 5, dtype=torch.float32).reshape(2, 8, 3, 5)
    assert np.array_equal(O.pixel_shuffle2(xg).numpy().astype(np.int32), arr["pixel_shuffle_out"])


def test_eight_wave_workgroup_shape_matches(tmp_path):
    """The eight-wave tile shapes (selected with SRX_CONV_WAVES=8, read once per process) give the
    same results as the default four-wave ones: run the generator block tests in a child process."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, SRX_CONV_WAVES="8")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                        os.path.join(here, "test_gpu_generator.py"), "-k", "blocks_match or generator_1block"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_stream_create_priorities_and_side_stream():
    """srx_stream_create: streams of low, normal and HIGH priority, each created late in the process's
    life (after other tests have run backward passes on the low-priority side stream), run the same
    convolution to the same bits as the main stream; a priority outside the device's range is an
    argument error; engine.side_stream hands out one stream per device.
    The new stream's launch is ordered after the main stream's set-up work by a HOST-side join: a
    device-side wait of a higher-priority stream for an event of a lower-priority one is the one
    pattern this runtime does not survive (with st.wait_stream(main) the priority -1 case left
    both the new stream and the main stream unfinished: hipStreamQuery 600 / 600, side stream 0).
    Completion is polled for a bounded time, and a stream that does not finish fails the test with
    every stream's hipStreamQuery state instead of blocking."""
    import ctypes as C
    import time
    E, L = pkg("engine"), pkg("_lib")
    lib = L.lib()
    dev = torch.device("cuda")
    x = O.hash_input("xs", (2, 64, 16, 32)).cuda()
    w = O.hash_uniform("ws", (32, 64, 3, 3), 0.05).cuda()
    b = O.hash_uniform("bs", (32,), 0.1).cuda()
    base = run_single_conv(x, w, b, tdtype=torch.bfloat16, act=0.2)
    hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    for prio in (1, 0, -1):
        # buffers, packed weights and the converted input on the main stream ...
        plan = E.Plan(dev, torch.bfloat16)
        X, Y = plan.buf(2, 64, 16, 32), plan.buf(2, 32, 16, 32, zero=True)
        wp, bp = plan.packed_layer(32, 64, 9, 0, [E.conv_seg(w)], b.data_ptr())
        plan.finalize_packing()
        plan.run_packing()
        plan.nchw_to_c8(x, (X, 0))
        ops = []
        plan.conv(ops, (X, 0), (Y, 0), wp, bp, 2, 16, 32, 64, 32, act=0.2)
        frozen = plan.freeze(ops)
        # ... the launch on a new stream of the given priority, after a host-side join
        sp = C.c_void_p()
        assert lib.srx_stream_create(prio, C.byref(sp)) == 0 and sp.value
        got = C.c_int(99)
        assert hip.hipStreamGetPriority(sp, C.byref(got)) == 0 and got.value == prio
        st = torch.cuda.ExternalStream(sp.value)
        torch.cuda.current_stream().synchronize()
        plan.run(frozen, stream=st)
        ev = torch.cuda.Event()
        ev.record(st)
        t0 = time.monotonic()
        while not ev.query() and time.monotonic() - t0 < 20.0:
            time.sleep(0.005)
        if not ev.query():
            q = lambda p: int(hip.hipStreamQuery(C.c_void_p(p)))
            pytest.fail("the convolution launched on a stream of priority %d did not finish in 20 s; hipStreamQuery "
                        "(0 = done, 600 = not ready): new stream %d, main stream %d, null stream %d, side stream %d"
                        % (prio, q(sp.value), q(torch.cuda.current_stream().cuda_stream), q(0),
                           q(E.side_stream(dev).cuda_stream)))
        out = plan.c8_to_nchw((Y, 0), 32).cpu()       # (the event has completed: a host-side join)
        assert torch.equal(out, base)
        assert lib.srx_stream_destroy(sp) == 0
    sp = C.c_void_p()
    assert lib.srx_stream_create(7, C.byref(sp)) != 0
    assert b"priority" in lib.srx_last_error()
    s1, s2 = E.side_stream(torch.device("cuda")), E.side_stream(torch.device("cuda", 0))
    assert s1 is s2


def test_flat_table_packing_equals_grid_packing():
    """srx_pack_weights_flat (one workgroup pass per (layer, unit) table entry) writes the same bits
    as srx_pack_weights_scaled (units x layers grid) for every packed tensor of a generator plan
    (forward layers, PixelShuffle row order, the stacked / transposed / tap-flipped data-gradient
    layers), of a discriminator plan (1/sigma scales read from the device, 1x1 layers) and for a
    sub-range of the layer table."""
    E, CH = pkg("engine"), pkg("chain")
    G, D = pkg("esrgan.generator"), pkg("esrgan.discriminator")
    from helpers import fill_module_
    g = fill_module_(G.GeneratorESRGAN(num_blocks=2), salt=41).cuda()
    d = fill_module_(D.Discriminator(), salt=42).cuda()
    gp = E.GeneratorPlan({n: p.data for n, p in g.named_parameters()}, 1, 16, 16, torch.bfloat16, num_blocks=2)
    dp = CH.DiscriminatorPlan({n: p.data for n, p in d.named_parameters()}, {n: b for n, b in d.named_buffers()},
                              1, 32, 32, torch.float32)
    dp.inv_sigma.copy_(torch.linspace(0.5, 1.5, dp.inv_sigma.numel()))

    def packed(plan, flat, first=0, count=None):
        for w, b in plan.packed_tensors:
            w.fill_(7.0)
            if b is not None:
                b.fill_(7.0)
        plan.flat_packing = flat
        plan.run_packing(first, count)
        torch.cuda.synchronize()
        return [(w.clone(), None if b is None else b.clone()) for w, b in plan.packed_tensors]

    for plan in (gp, dp):
        for first, count in ((0, None), (3, 5)):
            a, b = packed(plan, False, first, count), packed(plan, True, first, count)
            assert len(a) == len(plan.pack_layers) > 5
            for (wa, ba), (wb, bb) in zip(a, b):
                assert torch.equal(wa.view(torch.int16 if wa.dtype == torch.bfloat16 else torch.int32),
                                   wb.view(torch.int16 if wb.dtype == torch.bfloat16 else torch.int32))
                assert (ba is None) == (bb is None) and (ba is None or torch.equal(ba, bb))
        assert any(float(w.float().abs().max()) not in (0.0, 7.0) for w, _ in a)
