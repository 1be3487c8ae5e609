"""GPU parity of the inference engine: whole-image forward, the tile gather (bit-exact against
F.pad reflect + slicing), and the tiled blend against the oracle's restatement of
inference.py:187-230 run with the oracle generator on the CPU."""
import ctypes as C
import os

import pytest
import torch
import torch.nn.functional as F

from helpers import O, ROOT, fill_module_, golden, pkg, rel_err

pytestmark = pytest.mark.gpu


def _engine(num_blocks=1, half=False, window="hann", nf=64, **kw):
    G = pkg("esrgan.generator")
    I = pkg("inference")
    g = fill_module_(G.GeneratorESRGAN(num_features=nf, num_blocks=num_blocks), salt=11)
    eng = I.SuperResolutionInference("esrgan", model=g, half_precision=half, window=window, **kw)
    sd = {k: v.detach().cpu().clone() for k, v in g.state_dict().items()}
    return eng, sd


def test_whole_image_matches_oracle():
    eng, sd = _engine()
    x = (O.hash_input("inf.x", (2, 3, 20, 28)) + 1) / 2
    out = eng.upscale(x, clamp=False)
    ref = O.generator_forward(x, sd)
    assert out.shape == (2, 3, 80, 112)
    assert rel_err(out, ref) < 1e-3
    assert float(eng.upscale(x).max()) <= 1.0 and float(eng.upscale(x).min()) >= 0.0


def test_tile_gather_is_bit_exact():
    L = pkg("_lib")
    lib = L.lib()
    img = O.hash_input("inf.img", (3, 21, 30)).cuda().contiguous()
    h, w, th, tw = 21, 30, 8, 8
    padded = F.pad(img.unsqueeze(0), [0, 6, 0, 7], "reflect")[0]            # 28 x 36
    origins = [(0, 0), (20, 28), (12, 4), (20, 0)]
    tiles = (L.Tile * len(origins))()
    for t, (y, x) in zip(tiles, origins):
        t.y0, t.x0, t.th, t.tw, t.offset = y, x, th, tw, 0
    dev = torch.frombuffer(bytearray(C.string_at(C.addressof(tiles), C.sizeof(tiles))), dtype=torch.uint8).cuda()
    dst = torch.full((len(origins), th, tw, 8), 7.0, device="cuda")
    L.check(lib.srx_tile_gather(L.F32, img.data_ptr(), 3, h, w, dev.data_ptr(), len(origins), th, tw,
                                dst.data_ptr(), th * tw * 8, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    for i, (y, x) in enumerate(origins):
        want = padded[:, y:y + th, x:x + tw].permute(1, 2, 0)
        assert torch.equal(dst[i, :, :, :3], want)
        assert float(dst[i, :, :, 3:].abs().max()) == 0.0


@pytest.mark.parametrize("window", ["hann", "hann_pos"])
@pytest.mark.parametrize("shape,tile,overlap", [((1, 3, 40, 52), 16, 4), ((2, 3, 33, 20), 16, 0),
                                                ((1, 3, 24, 70), 24, 8)])
def test_tiled_matches_oracle(window, shape, tile, overlap):
    eng, sd = _engine(window=window, tile_batch=5)
    x = (O.hash_input("inf.t", shape) + 1) / 2
    out = eng.upscale(x, tile_size=tile, tile_overlap=overlap, clamp=False).cpu()
    ref = O.infer_tiled(x, lambda t: O.generator_forward(t, sd), tile, overlap, window=window)
    assert out.shape == ref.shape
    assert rel_err(out, ref) < 2e-3
    if window == "hann":       # the reference's window leaves the outermost rows/columns at weight 0
        assert float(out[:, :, 0, :].abs().max()) == 0.0
    else:
        assert float(out[:, :, 0, :].abs().max()) > 0.0


def test_tiled_bf16_close_and_u8_quantisation():
    eng32, sd = _engine(window="hann_pos")
    eng16, _ = _engine(half=True, window="hann_pos")
    x = (O.hash_input("inf.h", (1, 3, 40, 40)) + 1) / 2
    a = eng32.upscale(x, tile_size=16, tile_overlap=4)
    b = eng16.upscale(x, tile_size=16, tile_overlap=4)
    assert float((a - b).abs().max()) < 3e-2
    u8 = eng32._infer_tiled(x[0].cuda().contiguous(), 16, 4, True, want_u8=True)
    assert torch.equal(u8, a[0].mul(255).byte())


def test_small_image_skips_tiling_and_checkpoint_loads():
    man, arr = golden()
    case = man["cases"]["checkpoint_g"]
    I = pkg("inference")
    eng = I.SuperResolutionInference("esrgan", os.path.join(ROOT, "tests", "golden", case["file"]),
                                     num_features=case["num_features"], num_blocks=case["num_blocks"])
    x = O.hash_input("ckpt.x", (1, 3, 8, 8))
    out = eng.upscale(x, tile_size=16, clamp=False)          # 8 <= tile: whole-image path (inference.py:151)
    assert rel_err(out, torch.from_numpy(arr["ckpt_g_out"])) < 1e-3
    with pytest.raises(ValueError):
        I.SuperResolutionInference("swinir", model=eng.model)


def test_config4_tiling_as_stated_matches_oracle():
    """BASELINE config 4's tiling (tile 256, overlap 32; inference.py:187-230 defaults) on an image
    that needs reflect padding and clipped edge tiles: [1,3,300,330], one-block generator, fp32 MFMA,
    against the oracle's restatement of the tiler run with the oracle generator on the CPU."""
    eng, sd = _engine(window="hann_pos", tile_batch=4)
    x = (O.hash_input("inf.c4", (1, 3, 300, 330)) + 1) / 2
    out = eng.upscale(x, tile_size=256, tile_overlap=32, clamp=False).cpu()
    ref = O.infer_tiled(x, lambda t: O.generator_forward(t, sd), 256, 32, window="hann_pos")
    assert out.shape == ref.shape == (1, 3, 1200, 1320)
    assert rel_err(out, ref) < 2e-3


def test_config4_full_size_whole_image_properties():
    """Config 4's image size (512 x 512 LR -> 2048 x 2048), 23 blocks, bf16, whole image: finite,
    deterministic, and an image's output does not depend on the batch it is in (config 4 shards
    32 images over 8 GPUs: replicas only)."""
    eng, _ = _engine(num_blocks=23, half=True)
    x = (O.hash_input("inf.c4w", (2, 3, 512, 512)) + 1) / 2
    y = eng.upscale(x, clamp=False)
    assert y.shape == (2, 3, 2048, 2048)
    assert torch.isfinite(y).all() and float(y.abs().max()) > 0
    assert torch.equal(y, eng.upscale(x, clamp=False))
    assert torch.equal(eng.upscale(x[1:2], clamp=False), y[1:2])
