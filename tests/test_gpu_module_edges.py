"""Edge behaviour of the drop-in modules that callers of the reference's nn.Modules can hit:
empty batches, wrong channel counts, non-4-D inputs, CPU tensors."""
import pytest
import torch

from helpers import pkg

pytestmark = pytest.mark.gpu


def _mods():
    G, D = pkg("esrgan.generator"), pkg("esrgan.discriminator")
    return [
        (G.ResidualDenseBlock(64, 32).cuda(), (64, 8, 8), (64, 8, 8)),
        (G.RRDB(64).cuda(), (64, 8, 8), (64, 8, 8)),
        (G.UpsampleBlock(64).cuda(), (64, 8, 8), (64, 16, 16)),
        (G.GeneratorESRGAN(num_blocks=1).cuda(), (3, 8, 8), (3, 32, 32)),
    ]


def test_empty_batch_gives_empty_output_and_zero_grads():
    for m, cin, cout in _mods():
        x = torch.zeros((0,) + cin, device="cuda", requires_grad=True)
        y = m(x)
        assert tuple(y.shape) == (0,) + cout
        y.sum().backward()
        assert tuple(x.grad.shape) == (0,) + cin
        for p in m.parameters():
            assert p.grad is not None and float(p.grad.abs().sum()) == 0.0
        with torch.no_grad():
            assert tuple(m(x.detach()).shape) == (0,) + cout
    # the reference's Discriminator.forward ends in x.view(x.size(0), -1) (esrgan/discriminator.py:38),
    # which torch refuses for zero elements: same line, same error here
    d = pkg("esrgan.discriminator").Discriminator().cuda()
    with pytest.raises(RuntimeError, match="ambiguous"):
        d(torch.zeros((0, 3, 64, 64), device="cuda"))


def test_wrong_channels_and_rank_and_device_are_errors():
    for m, cin, _ in _mods():
        bad = torch.zeros((1, cin[0] + 8) + cin[1:], device="cuda")
        with pytest.raises(RuntimeError, match="channels"):
            m(bad)
        with pytest.raises(ValueError, match="4-D"):
            m(torch.zeros(cin, device="cuda"))
        with pytest.raises(RuntimeError, match="GPU"):
            m(torch.zeros((1,) + cin))
    d = pkg("esrgan.discriminator").Discriminator().cuda()
    with pytest.raises(RuntimeError, match="channels"):
        d(torch.zeros((1, 4, 64, 64), device="cuda"))
    with pytest.raises(ValueError, match="multiples of 16"):
        d(torch.zeros((1, 3, 40, 64), device="cuda"))
