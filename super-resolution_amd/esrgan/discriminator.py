"""VGG-style spectral-normalised discriminator with the reference's module interface
(reference ``esrgan/discriminator.py:5-38``), executed by libsrx.

The constructor builds exactly the reference's module tree with
``torch.nn.utils.spectral_norm`` so that parameter / buffer names
(``*.bias``, ``*.weight_orig``, ``*.weight_u``, ``*.weight_v``), their registration
order, the state_dict ``_metadata`` entry and the constructor's RNG consumption are
identical.  ``forward`` does not call those children (so their pre-forward hooks never
run): the power iteration, sigma, the normalised-weight packing and every convolution
run as HIP kernels on the same tensors, updating ``weight_u`` / ``weight_v`` in place in
training mode exactly as the hook does (torch spectral_norm.py:97-114).
"""
from __future__ import annotations

from torch import nn
from torch.nn.utils import spectral_norm

from .. import chain
from ..host import _SrxHost


# (output channels, stride) of the eight 3x3 stages, in order (reference esrgan/discriminator.py:18-26)
_STAGES = ((64, 1), (64, 2), (128, 1), (128, 2), (256, 1), (256, 2), (512, 1), (512, 2))
_HEAD_WIDTH = 1024


class Discriminator(_SrxHost):
    """Eight spectral-normalised 3x3 stages (four of them stride 2), global average pooling and a
    two-layer 1x1 head producing one logit per image.

    Module tree: ``features.{i}`` = Sequential(SN conv, LeakyReLU) for each stage,
    ``classifier`` = Sequential(pool, SN conv, LeakyReLU, SN conv) — the key layout of the
    reference's checkpoints.  Layers are created first to last, so the default initialisers and
    the spectral-norm u/v draws consume the RNG in the reference's order."""

    def __init__(self, in_channels=3):
        super().__init__()
        stages, width_in = [], in_channels
        for width_out, stride in _STAGES:
            conv = spectral_norm(nn.Conv2d(width_in, width_out, kernel_size=3, stride=stride, padding=1))
            stages.append(nn.Sequential(conv, nn.LeakyReLU(0.2, inplace=True)))
            width_in = width_out
        self.features = nn.Sequential(*stages)
        head = [nn.AdaptiveAvgPool2d(1)]
        head.append(spectral_norm(nn.Conv2d(width_in, _HEAD_WIDTH, kernel_size=1)))
        head.append(nn.LeakyReLU(0.2, inplace=True))
        head.append(spectral_norm(nn.Conv2d(_HEAD_WIDTH, 1, kernel_size=1)))
        self.classifier = nn.Sequential(*head)

    def _build_plan(self, params, B, h, w, tdtype, need_grad, need_input_grad):
        if h % 16 or w % 16:
            raise ValueError("Discriminator input height/width must be multiples of 16 "
                             f"(four stride-2 stages); got {h}x{w}")
        buffers = {n: b for n, b in self.named_buffers()}
        return chain.DiscriminatorPlan(params, buffers, B, h, w, tdtype,
                                       need_param_grad=need_grad, need_input_grad=need_input_grad)

    def _out_shape(self, shape):
        self._check_channels(shape, self.features[0][0].in_channels)
        return (shape[0], 1, 1, 1)

    def forward(self, x):
        x = self._run(x)
        return x.view(x.size(0), -1)
