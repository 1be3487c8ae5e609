"""CPU oracle for the ESRGAN hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain functional PyTorch on the CPU, the arithmetic of
the reference's ESRGAN generator, spectral-normalised discriminator, VGG-feature
loss and one adversarial training step.  It exists so the HIP path can be
checked against it; nothing in the product package imports it.  Only tests,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may.

Every function cites the reference lines it follows.  The restatement is pinned
against the real reference modules by ``tests/golden/make_golden.py`` (run in
the container that holds the reference), whose outputs are the committed
fixtures under ``tests/golden/``.  The VGG19 feature network is restated from
the public torchvision architecture (cfg "E"); the reference's pretrained
ImageNet weights are not available offline, so that piece is pinned only by
its own fixtures with synthetic weights ("parity unpinned" by the reference).

All tensors are NCHW; ``sd`` arguments are ``state_dict``-style mappings whose
keys are exactly the reference's.
"""
from __future__ import annotations

import zlib
from typing import Dict, List, Mapping, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Mapping[str, Tensor]

LRELU_SLOPE = 0.2          # esrgan/generator.py:15,45,63 ; esrgan/discriminator.py:13,31
RESIDUAL_SCALE = 0.2       # esrgan/generator.py:23,37


# --------------------------------------------------------------------------
# deterministic parameter filler (independent of torch's RNG)
# --------------------------------------------------------------------------
def hash_uniform(name: str, shape, scale: float = 1.0, salt: int = 0) -> Tensor:
    """U(-scale, scale) values that depend only on (name, flat index, salt).

    splitmix64 of ``crc32(name) * 2^32 + index``; the top 24 bits become the
    mantissa so that the value is exactly representable in fp32.
    """
    n = int(np.prod(shape)) if len(shape) else 1
    seed = np.uint64(zlib.crc32(name.encode()) ^ (salt * 0x9E3779B1 & 0xFFFFFFFF))
    with np.errstate(over="ignore"):
        z = (seed << np.uint64(32)) + np.arange(n, dtype=np.uint64)
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(40)).astype(np.float64) / float(1 << 24)      # [0,1)
    v = ((u * 2.0 - 1.0) * scale).astype(np.float32)
    return torch.from_numpy(v.reshape(tuple(shape)))


def fill_state_dict_(sd: Dict[str, Tensor], salt: int = 0, gain: float = 1.0) -> None:
    """Overwrite every tensor of ``sd`` in place with hash_uniform values.

    Weights (ndim>1) get scale gain*sqrt(3/fan_in) (unit-variance-preserving),
    biases 0.1, spectral-norm u/v vectors are filled then L2-normalised (as
    torch/nn/utils/spectral_norm.py:168-169 does for its random init).
    """
    for k, t in sd.items():
        if not torch.is_floating_point(t):
            continue
        if k.endswith("weight_u") or k.endswith("weight_v"):
            val = hash_uniform(k, t.shape, 1.0, salt)
            val = F.normalize(val, dim=0, eps=1e-12)
        elif t.ndim > 1:
            fan_in = int(np.prod(t.shape[1:]))
            val = hash_uniform(k, t.shape, gain * (3.0 / fan_in) ** 0.5, salt)
        else:
            val = hash_uniform(k, t.shape, 0.1, salt)
        with torch.no_grad():
            t.copy_(val.to(t.dtype))


# --------------------------------------------------------------------------
# generator  (esrgan/generator.py)
# --------------------------------------------------------------------------
def lrelu(x: Tensor) -> Tensor:
    return torch.where(x > 0, x, x * LRELU_SLOPE)


def conv3x3(x: Tensor, w: Tensor, b: Optional[Tensor], stride: int = 1) -> Tensor:
    """nn.Conv2d(cin, cout, 3, stride, 1) — esrgan/generator.py:10-14."""
    return F.conv2d(x, w, b, stride=stride, padding=1)


def pixel_shuffle2(x: Tensor) -> Tensor:
    """nn.PixelShuffle(2) (esrgan/generator.py:44), restated as pure indexing:
    out[n, c, 2y+i, 2x+j] = in[n, 4c + 2i + j, y, x]."""
    n, c4, h, w = x.shape
    c = c4 // 4
    x = x.reshape(n, c, 2, 2, h, w)            # [n, c, i, j, y, x]
    x = x.permute(0, 1, 4, 2, 5, 3)            # [n, c, y, i, x, j]
    return x.reshape(n, c, 2 * h, 2 * w)


def rdb_forward(x: Tensor, sd: SD, prefix: str = "") -> Tensor:
    """ResidualDenseBlock.forward — esrgan/generator.py:17-23."""
    p = prefix
    x1 = lrelu(conv3x3(x, sd[p + "conv1.weight"], sd[p + "conv1.bias"]))
    x2 = lrelu(conv3x3(torch.cat((x, x1), 1), sd[p + "conv2.weight"], sd[p + "conv2.bias"]))
    x3 = lrelu(conv3x3(torch.cat((x, x1, x2), 1), sd[p + "conv3.weight"], sd[p + "conv3.bias"]))
    x4 = lrelu(conv3x3(torch.cat((x, x1, x2, x3), 1), sd[p + "conv4.weight"], sd[p + "conv4.bias"]))
    x5 = conv3x3(torch.cat((x, x1, x2, x3, x4), 1), sd[p + "conv5.weight"], sd[p + "conv5.bias"])
    return x5 * RESIDUAL_SCALE + x


def rrdb_forward(x: Tensor, sd: SD, prefix: str = "") -> Tensor:
    """RRDB.forward — esrgan/generator.py:33-37."""
    out = rdb_forward(x, sd, prefix + "rdb1.")
    out = rdb_forward(out, sd, prefix + "rdb2.")
    out = rdb_forward(out, sd, prefix + "rdb3.")
    return out * RESIDUAL_SCALE + x


def upsample_forward(x: Tensor, sd: SD, prefix: str = "") -> Tensor:
    """UpsampleBlock.forward — esrgan/generator.py:47-48."""
    return lrelu(pixel_shuffle2(conv3x3(x, sd[prefix + "conv.weight"], sd[prefix + "conv.bias"])))


def generator_num_blocks(sd: SD) -> int:
    n = 0
    while f"body.{n}.rdb1.conv1.weight" in sd:
        n += 1
    return n


def generator_forward(x: Tensor, sd: SD) -> Tensor:
    """GeneratorESRGAN.forward — esrgan/generator.py:65-72 (no output activation)."""
    feat = conv3x3(x, sd["conv_first.weight"], sd["conv_first.bias"])
    body = feat
    for i in range(generator_num_blocks(sd)):
        body = rrdb_forward(body, sd, f"body.{i}.")
    trunk = conv3x3(body, sd["conv_body.weight"], sd["conv_body.bias"])
    feat = feat + trunk
    feat = upsample_forward(feat, sd, "upsample.0.")
    feat = upsample_forward(feat, sd, "upsample.1.")
    out = conv3x3(lrelu(conv3x3(feat, sd["conv_hr.weight"], sd["conv_hr.bias"])),
                  sd["conv_last.weight"], sd["conv_last.bias"])
    return out


def generator_state_dict_spec(num_blocks: int = 23, in_ch: int = 3, out_ch: int = 3,
                              nf: int = 64, gc: int = 32) -> List[Tuple[str, Tuple[int, ...]]]:
    """(key, shape) list in the reference's registration order
    (esrgan/generator.py:10-14,43,54-62)."""
    spec: List[Tuple[str, Tuple[int, ...]]] = []

    def conv(name, cin, cout):
        spec.append((name + ".weight", (cout, cin, 3, 3)))
        spec.append((name + ".bias", (cout,)))

    conv("conv_first", in_ch, nf)
    for b in range(num_blocks):
        for r in (1, 2, 3):
            for k in range(1, 5):
                conv(f"body.{b}.rdb{r}.conv{k}", nf + (k - 1) * gc, gc)
            conv(f"body.{b}.rdb{r}.conv5", nf + 4 * gc, nf)
    conv("conv_body", nf, nf)
    conv("upsample.0.conv", nf, nf * 4)
    conv("upsample.1.conv", nf, nf * 4)
    conv("conv_hr", nf, nf)
    conv("conv_last", nf, out_ch)
    return spec


# --------------------------------------------------------------------------
# spectral norm + discriminator (esrgan/discriminator.py; torch spectral_norm.py)
# --------------------------------------------------------------------------
D_FEATURES = [(3, 64, 1), (64, 64, 2), (64, 128, 1), (128, 128, 2),
              (128, 256, 1), (256, 256, 2), (256, 512, 1), (512, 512, 2)]  # discriminator.py:18-25
D_LAYERS = [f"features.{i}.0" for i in range(8)] + ["classifier.1", "classifier.3"]


def discriminator_state_dict_spec(in_ch: int = 3) -> List[Tuple[str, Tuple[int, ...]]]:
    """(key, shape) in state_dict order: per layer bias, weight_orig (params)
    then weight_u, weight_v (buffers) — esrgan/discriminator.py:12,30,32 +
    torch/nn/utils/spectral_norm.py:171-180."""
    spec = []
    shapes = [(co, (in_ch if i == 0 else ci), 3, 3) for i, (ci, co, _) in enumerate(D_FEATURES)]
    shapes += [(1024, 512, 1, 1), (1, 1024, 1, 1)]
    for name, shp in zip(D_LAYERS, shapes):
        spec.append((name + ".bias", (shp[0],)))
        spec.append((name + ".weight_orig", shp))
        spec.append((name + ".weight_u", (shp[0],)))
        spec.append((name + ".weight_v", (shp[1] * shp[2] * shp[3],)))
    return spec


def sn_weight(w_orig: Tensor, u: Tensor, v: Tensor, do_power_iteration: bool,
              eps: float = 1e-12) -> Tuple[Tensor, Tensor, Tensor]:
    """SpectralNorm.compute_weight — torch/nn/utils/spectral_norm.py:92-114.

    Returns (w_sn, u_new, v_new).  One power iteration when training; u and v
    are constants for autograd; sigma = u . (W v) carries gradient to w_orig.
    """
    wm = w_orig.reshape(w_orig.shape[0], -1)
    if do_power_iteration:
        with torch.no_grad():
            v = F.normalize(torch.mv(wm.t(), u), dim=0, eps=eps)
            u = F.normalize(torch.mv(wm, v), dim=0, eps=eps)
    u = u.detach()
    v = v.detach()
    sigma = torch.dot(u, torch.mv(wm, v))
    return w_orig / sigma, u, v


def discriminator_forward(x: Tensor, sd: SD, training: bool = True,
                          state: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """Discriminator.forward — esrgan/discriminator.py:35-38.

    ``state`` (optional dict) holds the evolving u/v buffers; if None the ones
    in ``sd`` are used and not updated.  When given, it is updated in place the
    way the hook updates module buffers (spectral_norm.py:103-106).
    """
    def uv(name):
        src = state if state is not None else sd
        return src[name + ".weight_u"], src[name + ".weight_v"]

    def sn(name):
        u, v = uv(name)
        w, u2, v2 = sn_weight(sd[name + ".weight_orig"], u, v, training)
        if state is not None:
            state[name + ".weight_u"] = u2
            state[name + ".weight_v"] = v2
        return w

    for i, (_, _, stride) in enumerate(D_FEATURES):
        name = f"features.{i}.0"
        x = lrelu(conv3x3(x, sn(name), sd[name + ".bias"], stride))
    x = x.mean(dim=(2, 3), keepdim=True)                     # AdaptiveAvgPool2d(1), :29
    x = lrelu(F.conv2d(x, sn("classifier.1"), sd["classifier.1.bias"]))
    x = F.conv2d(x, sn("classifier.3"), sd["classifier.3.bias"])
    return x.reshape(x.shape[0], -1)


# --------------------------------------------------------------------------
# VGG19 features[:36] and the perceptual loss (utils/loss.py:8-22)
# --------------------------------------------------------------------------
VGG19_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M",
             512, 512, 512, 512, "M", 512, 512, 512, 512]     # torchvision cfg "E" minus last pool
VGG19_CONV_IDX = [0, 2, 5, 7, 10, 12, 14, 16, 19, 21, 23, 25, 28, 30, 32, 34]


def vgg19_state_dict_spec() -> List[Tuple[str, Tuple[int, ...]]]:
    spec, cin, it = [], 3, iter(VGG19_CONV_IDX)
    for c in VGG19_CFG:
        if c == "M":
            continue
        i = next(it)
        spec.append((f"{i}.weight", (c, cin, 3, 3)))
        spec.append((f"{i}.bias", (c,)))
        cin = c
    return spec


def vgg19_features36(x: Tensor, sd: SD) -> Tensor:
    """torchvision vgg19().features[:36]: conv3x3+ReLU pairs and four 2x2
    max-pools, ending at relu5_4 (utils/loss.py:15)."""
    it = iter(VGG19_CONV_IDX)
    for c in VGG19_CFG:
        if c == "M":
            x = F.max_pool2d(x, 2, 2)
        else:
            i = next(it)
            x = F.relu(conv3x3(x, sd[f"{i}.weight"], sd[f"{i}.bias"]))
    return x


def vgg_loss(fake: Tensor, real: Tensor, vgg_sd: SD) -> Tensor:
    """VGGLoss.forward — utils/loss.py:21-22 (L1 of features, no input normalisation)."""
    return (vgg19_features36(fake, vgg_sd) - vgg19_features36(real, vgg_sd)).abs().mean()


# --------------------------------------------------------------------------
# losses and the adversarial step (esrgan/train_esrgan.py:87-111)
# --------------------------------------------------------------------------
def bce_with_logits(logits: Tensor, target_value: float) -> Tensor:
    """nn.BCEWithLogitsLoss() against a constant target, mean reduction
    (esrgan/train_esrgan.py:91-92,103)."""
    t = torch.full_like(logits, target_value)
    return (torch.clamp(logits, min=0) - logits * t + torch.log1p(torch.exp(-logits.abs()))).mean()


def l1_loss(a: Tensor, b: Tensor) -> Tensor:
    return (a - b).abs().mean()


def adversarial_step_losses(lr: Tensor, hr: Tensor, g_sd: SD, d_sd: SD, vgg_sd: SD,
                            d_state: Dict[str, Tensor],
                            lambda_l1: float = 1.0, lambda_adv: float = 5e-3,
                            lambda_percep: float = 1.0, d_sd_after_step: Optional[SD] = None):
    """One D-step + G-step's loss graph on fp32 CPU (autocast disabled, as the
    reference does when DEVICE != 'cuda': esrgan/train_esrgan.py:87,101).

    Returns dict of scalars: loss_disc, gen_adv_loss, gen_l1_loss, gen_vgg_loss,
    loss_gen, plus 'fake'.  The three discriminator forwards each advance the
    power iteration (state is updated in place).  ``d_sd_after_step`` lets the
    caller supply the discriminator weights as updated by opt_d.step() between
    the two phases (train_esrgan.py:97,102); defaults to the same weights.
    """
    fake = generator_forward(lr, g_sd)                                          # :88
    disc_real = discriminator_forward(hr, d_sd, True, d_state)                  # :89
    disc_fake = discriminator_forward(fake.detach(), d_sd, True, d_state)       # :90
    loss_disc = (bce_with_logits(disc_real, 1.0) + bce_with_logits(disc_fake, 0.0)) / 2   # :91-93
    d2 = d_sd_after_step if d_sd_after_step is not None else d_sd
    disc_fake_for_gen = discriminator_forward(fake, d2, True, d_state)          # :102
    gen_adv = bce_with_logits(disc_fake_for_gen, 1.0)                           # :103
    gen_l1 = l1_loss(fake, hr)                                                  # :104
    gen_vgg = vgg_loss(fake, hr, vgg_sd)                                        # :105
    loss_gen = lambda_l1 * gen_l1 + lambda_adv * gen_adv + lambda_percep * gen_vgg   # :106
    return {"fake": fake, "loss_disc": loss_disc, "gen_adv_loss": gen_adv,
            "gen_l1_loss": gen_l1, "gen_vgg_loss": gen_vgg, "loss_gen": loss_gen}


def psnr_unit_peak(sr: Tensor, hr: Tensor) -> Tensor:
    """10*log10(1/mse) — esrgan/train_esrgan.py:59-60."""
    return 10 * torch.log10(1 / F.mse_loss(sr, hr))


# --------------------------------------------------------------------------
# helpers used by the tests / bench
# --------------------------------------------------------------------------
def make_state_dict(spec, salt: int = 0, dtype=torch.float32, gain: float = 1.0) -> Dict[str, Tensor]:
    sd = {k: torch.empty(shp, dtype=dtype) for k, shp in spec}
    fill_state_dict_(sd, salt, gain)
    return sd


def hash_input(name: str, shape, salt: int = 0) -> Tensor:
    """U(-1,1) synthetic image batch (value range of utils/datasets.py:38)."""
    return hash_uniform(name, shape, 1.0, salt)


# ---------------------------------------------------------------------------
# Tiled inference (reference inference.py:187-230, intended semantics)
# ---------------------------------------------------------------------------
def blend_window(n: int, kind: str = "hann") -> Tensor:
    """1-D blending window of n samples.  "hann": torch.hann_window(n, periodic=False), what
    inference.py:208-209 builds (zero at both ends).  "hann_pos": the interior n samples of an
    (n+2)-point Hann window, strictly positive, so image borders and overlap-0 seams keep a
    non-zero weight."""
    if kind == "hann":
        return torch.hann_window(n, periodic=False, dtype=torch.float32)
    if kind == "hann_pos":
        return torch.hann_window(n + 2, periodic=False, dtype=torch.float32)[1:-1].clone()
    raise ValueError(kind)


def tile_grid(h: int, w: int, tile: int, overlap: int):
    """Padded size and tile origins of inference.py:191-199,214-217: returns
    (ph, pw, [(y, x, th, tw)] in the reference's loop order)."""
    stride = tile - overlap
    pad_h = max(0, (stride - (h - tile) % stride) % stride)
    pad_w = max(0, (stride - (w - tile) % stride) % stride)
    ph, pw = h + pad_h, w + pad_w
    tiles = []
    for y in range(0, ph, stride):
        for x in range(0, pw, stride):
            tiles.append((y, x, min(y + tile, ph) - y, min(x + tile, pw) - x))
    return ph, pw, tiles


def infer_tiled(lr: Tensor, model_fn, tile: int, overlap: int, scale: int = 4,
                window: str = "hann") -> Tensor:
    """inference.py:187-230 with the window sliced on its spatial dims for clipped edge tiles
    (the reference slices dims 0/1 of a 4-D tensor at :225-226, which cannot broadcast).
    model_fn: [b,c,th,tw] -> [b,c,th*scale,tw*scale]."""
    b, c, h, w = lr.shape
    ph, pw, tiles = tile_grid(h, w, tile, overlap)
    lr_padded = F.pad(lr, [0, pw - w, 0, ph - h], "reflect")
    sr = torch.zeros((b, c, ph * scale, pw * scale), dtype=torch.float32)
    wsum = torch.zeros_like(sr)
    w1 = blend_window(tile * scale, window)
    win = (w1.unsqueeze(1) * w1.unsqueeze(0)).expand(1, c, -1, -1)
    for (y, x, th, tw) in tiles:
        sr_tile = model_fn(lr_padded[:, :, y:y + th, x:x + tw]).float()
        ys, xs = y * scale, x * scale
        sh, sw = sr_tile.shape[2], sr_tile.shape[3]
        sr[:, :, ys:ys + sh, xs:xs + sw] += sr_tile * win[:, :, :sh, :sw]
        wsum[:, :, ys:ys + sh, xs:xs + sw] += win[:, :, :sh, :sw]
    out = sr / (wsum + 1e-8)
    return out[:, :, :h * scale, :w * scale]


# ---------------------------------------------------------------------------
# Training-pair augmentation (reference utils/datasets.py:77-114, after decoding)
# ---------------------------------------------------------------------------
def crop_augment(img_u8: np.ndarray, y0: int, x0: int, size: int, hflip: bool, vflip: bool,
                 quarter_turns: int) -> Tensor:
    """TF.crop -> TF.hflip -> TF.vflip -> TF.rotate(90*k, counter-clockwise, square patch) ->
    ToTensor -> Normalize(0.5, 0.5) on a uint8 [H,W,3] array; returns fp32 [3,size,size]."""
    p = img_u8[y0:y0 + size, x0:x0 + size]
    if hflip:
        p = p[:, ::-1]
    if vflip:
        p = p[::-1]
    p = np.rot90(p, quarter_turns, axes=(0, 1))
    t = torch.from_numpy(np.ascontiguousarray(p)).permute(2, 0, 1).to(torch.float32).div(255)
    return t.sub(0.5).div(0.5)
