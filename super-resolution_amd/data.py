"""Device-side batch former for training pairs: the work of the reference's
``TrainDataset.__getitem__`` (``utils/datasets.py:77-114``) after image decoding, for a whole
batch in two launches (LR, HR).  Decoded uint8 images stay resident in HBM; the host only
draws the random decisions (crop origin, flips, rotation), so feeding eight GPUs does not
depend on DataLoader worker processes doing per-pixel work.
"""
from __future__ import annotations

import ctypes as C
import random
from typing import List, Sequence, Tuple

import torch

from . import _lib as L
from . import engine

SCALE = 4


def draw_params(hr_h: int, hr_w: int, hr_crop: int, rng: random.Random) -> Tuple[int, int, int]:
    """The random decisions of utils/datasets.py:83-108 in the same order: crop origin (i, j)
    on the HR image, hflip, vflip, rotate?+angle.  Returns (i, j, flags) with flags bit 0 hflip,
    bit 1 vflip, bits 2-3 the number of counter-clockwise quarter turns."""
    i = rng.randint(0, hr_h - hr_crop)
    j = rng.randint(0, hr_w - hr_crop)
    flags = 0
    if rng.random() > 0.5:
        flags |= 1
    if rng.random() > 0.5:
        flags |= 2
    if rng.random() > 0.5:
        flags |= (rng.choice([90, 180, 270]) // 90) << 2
    return i, j, flags


class DeviceBatcher:
    """lr_images / hr_images: lists of uint8 [H,W,3] tensors on the device (decoded pairs)."""

    def __init__(self, lr_images: Sequence[torch.Tensor], hr_images: Sequence[torch.Tensor], hr_crop: int = 128,
                 scale: int = SCALE):
        if len(lr_images) != len(hr_images) or not lr_images:
            raise ValueError("need the same, non-zero number of LR and HR images")
        for a in list(lr_images) + list(hr_images):
            if a.dtype != torch.uint8 or a.dim() != 3 or a.shape[2] != 3 or not a.is_cuda or not a.is_contiguous():
                raise ValueError("images must be contiguous uint8 [H,W,3] tensors on the device")
        self.lr, self.hr = list(lr_images), list(hr_images)
        self.hr_crop, self.lr_crop, self.scale = hr_crop, hr_crop // scale, scale
        self.lib = L.lib()
        self._keep: List[torch.Tensor] = []

    def _launch(self, imgs, entries, size, out):
        arr = (L.Crop * len(entries))()
        for e, (idx, y0, x0, flags) in zip(arr, entries):
            img = imgs[idx]
            H, W = img.shape[0], img.shape[1]
            if y0 < 0 or x0 < 0 or y0 + size > H or x0 + size > W:
                raise ValueError(f"crop ({y0},{x0})+{size} leaves image {idx} of size {H}x{W}")
            e.img, e.H, e.W, e.y0, e.x0, e.flags = img.data_ptr(), H, W, y0, x0, flags
        dev = torch.frombuffer(bytearray(C.string_at(C.addressof(arr), C.sizeof(arr))), dtype=torch.uint8).to(out.device)
        self._keep.append(dev)
        L.check(self.lib.srx_crop_augment(dev.data_ptr(), len(entries), size, out.data_ptr(),
                                          engine.stream_ptr(out.device)), "srx_crop_augment")

    def batch(self, samples: Sequence[Tuple[int, int, int, int]]) -> Tuple[torch.Tensor, torch.Tensor]:
        """samples: (image index, HR crop row i, HR crop column j, flags) per batch entry.
        Returns (lr [B,3,p,p], hr [B,3,4p,4p]) fp32 in [-1,1]."""
        dev = self.lr[0].device
        B = len(samples)
        self._keep = self._keep[-4:]
        lr = torch.empty((B, 3, self.lr_crop, self.lr_crop), dtype=torch.float32, device=dev)
        hr = torch.empty((B, 3, self.hr_crop, self.hr_crop), dtype=torch.float32, device=dev)
        s = self.scale
        self._launch(self.hr, [(k, i, j, f) for k, i, j, f in samples], self.hr_crop, hr)
        self._launch(self.lr, [(k, i // s, j // s, f) for k, i, j, f in samples], self.lr_crop, lr)   # :92-96
        return lr, hr

    def random_batch(self, batch_size: int, rng: random.Random) -> Tuple[torch.Tensor, torch.Tensor]:
        samples = []
        for _ in range(batch_size):
            k = rng.randrange(len(self.hr))
            i, j, f = draw_params(self.hr[k].shape[0], self.hr[k].shape[1], self.hr_crop, rng)
            samples.append((k, i, j, f))
        return self.batch(samples)
