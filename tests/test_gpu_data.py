"""Device-side batch former against the oracle's restatement of utils/datasets.py:77-114
(bit-exact: integer indexing plus two correctly rounded fp32 divisions).  torchvision is not
installed, so this is pinned by the oracle only."""
import random

import numpy as np
import pytest
import torch

from helpers import O, pkg

pytestmark = pytest.mark.gpu


def test_crop_flip_rotate_normalise_bit_exact():
    Dm = pkg("data")
    rs = np.random.RandomState(3)
    hr_imgs = [rs.randint(0, 256, size=(96 + 8 * k, 120 - 4 * k, 3), dtype=np.uint8) for k in range(3)]
    lr_imgs = [rs.randint(0, 256, size=(h.shape[0] // 4, h.shape[1] // 4, 3), dtype=np.uint8) for h in hr_imgs]
    b = Dm.DeviceBatcher([torch.from_numpy(a).cuda() for a in lr_imgs],
                         [torch.from_numpy(a).cuda() for a in hr_imgs], hr_crop=32)
    samples = [(k % 3, 4 * (k % 5), 8 * (k % 3) + 3, f) for k, f in
               enumerate([0, 1, 2, 3, 4, 8, 12, 5, 10, 15, 7, 14])]
    lr, hr = b.batch(samples)
    torch.cuda.synchronize()
    assert lr.shape == (12, 3, 8, 8) and hr.shape == (12, 3, 32, 32)
    for n, (k, i, j, f) in enumerate(samples):
        want_hr = O.crop_augment(hr_imgs[k], i, j, 32, bool(f & 1), bool(f & 2), (f >> 2) & 3)
        want_lr = O.crop_augment(lr_imgs[k], i // 4, j // 4, 8, bool(f & 1), bool(f & 2), (f >> 2) & 3)
        assert torch.equal(hr[n].cpu(), want_hr), (n, f)
        assert torch.equal(lr[n].cpu(), want_lr), (n, f)
    assert float(hr.min()) >= -1.0 and float(hr.max()) <= 1.0
    with pytest.raises(ValueError):
        b.batch([(0, 90, 0, 0)])                     # crop leaves the image


def test_random_batch_draws_valid_samples():
    Dm = pkg("data")
    rng = random.Random(5)
    for _ in range(200):
        i, j, f = Dm.draw_params(100, 140, 64, rng)
        assert 0 <= i <= 36 and 0 <= j <= 76 and 0 <= f < 16
    hr = [torch.randint(0, 256, (64, 64, 3), dtype=torch.uint8).cuda()]
    lr = [torch.randint(0, 256, (16, 16, 3), dtype=torch.uint8).cuda()]
    a, bb = Dm.DeviceBatcher(lr, hr, hr_crop=32).random_batch(5, rng)
    assert a.shape == (5, 3, 8, 8) and bb.shape == (5, 3, 32, 32) and torch.isfinite(bb).all()
