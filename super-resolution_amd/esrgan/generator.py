"""RRDBNet generator with the reference's module interface, executed by libsrx.

Same class names, constructor arguments, parameter names / registration order
and forward signatures as the reference's ``esrgan/generator.py`` (lines 6-72),
so ``state_dict()`` / ``load_state_dict(strict=True)`` / ``optim.Adam(
gen.parameters())`` / ``.to()`` / ``.train()`` / ``.eval()`` behave identically
and existing checkpoints load unchanged.  The ``nn.Conv2d`` children are kept
as parameter containers only: ``forward`` never calls them; it runs a recorded
HIP op list (``engine.GeneratorPlan`` / ``engine.BlockPlan``) on the module's
parameters.  There is no CPU fallback.
"""
from __future__ import annotations

import torch
from torch import nn

from .. import engine


from ..host import _SrxHost  # noqa: E402


class ResidualDenseBlock(_SrxHost):
    """Innermost block of the RRDB: five densely connected 3x3 convolutions
    (reference esrgan/generator.py:6-23)."""

    def __init__(self, in_channels=64, growth_channels=32):
        super().__init__()
        # conv k reads the block input and the k-1 growth slices before it; the fifth maps back to
        # the block width.  Registered as conv1 .. conv5 (the checkpoint key names), in that order.
        for k in range(1, 6):
            width_in = in_channels + (k - 1) * growth_channels
            width_out = growth_channels if k < 5 else in_channels
            setattr(self, f"conv{k}", nn.Conv2d(width_in, width_out, kernel_size=3, stride=1, padding=1))
        self.lrelu = nn.LeakyReLU(0.2, inplace=True)

    def _build_plan(self, params, B, h, w, tdtype, need_grad, need_input_grad):
        nf, gc = self.conv1.in_channels, self.conv1.out_channels
        return engine.BlockPlan("rdb", params, B, h, w, tdtype, nf=nf, gc=gc,
                                need_grad=need_grad, need_input_grad=need_input_grad)

    def _out_shape(self, shape):
        self._check_channels(shape, self.conv1.in_channels)
        return shape

    def forward(self, x):
        return self._run(x)


class RRDB(_SrxHost):
    """Residual-in-residual dense block (reference esrgan/generator.py:25-37)."""

    def __init__(self, in_channels=64):
        super().__init__()
        self.rdb1 = ResidualDenseBlock(in_channels)
        self.rdb2 = ResidualDenseBlock(in_channels)
        self.rdb3 = ResidualDenseBlock(in_channels)

    def _build_plan(self, params, B, h, w, tdtype, need_grad, need_input_grad):
        nf, gc = self.rdb1.conv1.in_channels, self.rdb1.conv1.out_channels
        return engine.BlockPlan("rrdb", params, B, h, w, tdtype, nf=nf, gc=gc,
                                need_grad=need_grad, need_input_grad=need_input_grad)

    def _out_shape(self, shape):
        self._check_channels(shape, self.rdb1.conv1.in_channels)
        return shape

    def forward(self, x):
        return self._run(x)


class UpsampleBlock(_SrxHost):
    """conv 3x3 (C -> 4C) + PixelShuffle(2) + LeakyReLU(0.2)
    (reference esrgan/generator.py:39-48)."""

    def __init__(self, in_channels):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, in_channels * 4, 3, 1, 1)
        self.pixel_shuffle = nn.PixelShuffle(2)
        self.lrelu = nn.LeakyReLU(0.2, inplace=True)

    def _build_plan(self, params, B, h, w, tdtype, need_grad, need_input_grad):
        return engine.BlockPlan("upsample", params, B, h, w, tdtype, nf=self.conv.in_channels, gc=32,
                                need_grad=need_grad, need_input_grad=need_input_grad)

    def _out_shape(self, shape):
        self._check_channels(shape, self.conv.in_channels)
        return (shape[0], shape[1], 2 * shape[2], 2 * shape[3])

    def forward(self, x):
        return self._run(x)


class GeneratorESRGAN(_SrxHost):
    """The complete 4x generator (reference esrgan/generator.py:50-72)."""

    def __init__(self, in_channels=3, out_channels=3, num_features=64, num_blocks=23):
        super().__init__()
        nf = num_features

        def conv3(width_in, width_out):
            return nn.Conv2d(width_in, width_out, kernel_size=3, stride=1, padding=1)

        # registration order = parameters() order = the reference's state_dict order:
        # conv_first, body.{i}, conv_body, upsample.{0,1}, conv_hr, conv_last
        self.conv_first = conv3(in_channels, nf)
        self.body = nn.Sequential(*(RRDB(nf) for _ in range(num_blocks)))
        self.conv_body = conv3(nf, nf)
        self.upsample = nn.Sequential(*(UpsampleBlock(nf) for _ in range(2)))      # 2 x 2 = the 4x scale
        self.conv_hr = conv3(nf, nf)
        self.conv_last = conv3(nf, out_channels)
        self.lrelu = nn.LeakyReLU(0.2, inplace=True)

    def _build_plan(self, params, B, h, w, tdtype, need_grad, need_input_grad):
        gc = self.body[0].rdb1.conv1.out_channels if len(self.body) else 32
        return engine.GeneratorPlan(params, B, h, w, tdtype,
                                    in_ch=self.conv_first.in_channels, out_ch=self.conv_last.out_channels,
                                    nf=self.conv_first.out_channels, gc=gc, num_blocks=len(self.body),
                                    need_grad=need_grad, need_input_grad=need_input_grad)

    def _out_shape(self, shape):
        self._check_channels(shape, self.conv_first.in_channels)
        return (shape[0], self.conv_last.out_channels, 4 * shape[2], 4 * shape[3])

    def forward(self, x):
        return self._run(x)
