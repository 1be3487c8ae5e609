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


class Discriminator(_SrxHost):
    """A VGG-style discriminator with spectral normalization."""

    def __init__(self, in_channels=3):
        super().__init__()

        def conv_block(in_feat, out_feat, stride=1):
            return nn.Sequential(
                spectral_norm(nn.Conv2d(in_feat, out_feat, 3, stride=stride, padding=1)),
                nn.LeakyReLU(0.2, inplace=True)
            )

        self.features = nn.Sequential(
            conv_block(in_channels, 64),
            conv_block(64, 64, stride=2),
            conv_block(64, 128),
            conv_block(128, 128, stride=2),
            conv_block(128, 256),
            conv_block(256, 256, stride=2),
            conv_block(256, 512),
            conv_block(512, 512, stride=2),
        )

        self.classifier = nn.Sequential(
            nn.AdaptiveAvgPool2d(1),
            spectral_norm(nn.Conv2d(512, 1024, 1)),
            nn.LeakyReLU(0.2, inplace=True),
            spectral_norm(nn.Conv2d(1024, 1, 1))
        )

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
