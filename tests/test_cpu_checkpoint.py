"""Checkpoint formats of the reference (esrgan/train_esrgan.py:143-207, utils/checkpointing.py,
inference.py:133-141) written and read by the package's helpers; FlatAdam state in
torch.optim.Adam's layout.  Runs without a GPU: only state handling is exercised."""
import os

import torch
from torch import nn

from helpers import O, ROOT, fill_module_, golden, pkg


def _small_models():
    G = pkg("esrgan.generator")
    D = pkg("esrgan.discriminator")
    g = fill_module_(G.GeneratorESRGAN(num_features=16, num_blocks=1), salt=21)
    d = fill_module_(D.Discriminator(), salt=22)
    return g, d


def test_flat_adam_state_dict_is_torch_adam_layout():
    T = pkg("train")
    F = pkg("utils.flat")
    torch.manual_seed(0)
    m1 = nn.Sequential(nn.Linear(5, 4), nn.Linear(4, 3))
    opt = torch.optim.Adam(m1.parameters(), lr=2e-4, betas=(0.9, 0.999))
    for _ in range(3):
        opt.zero_grad()
        m1(torch.randn(7, 5)).square().mean().backward()
        opt.step()
    m2 = nn.Sequential(nn.Linear(5, 4), nn.Linear(4, 3))
    m2.load_state_dict(m1.state_dict())
    flat = F.flatten_parameters(m2)
    fa = T.FlatAdam(flat, lr=1.0, shapes=[p.shape for p in m2.parameters()])
    fa.load_state_dict(opt.state_dict())
    assert fa.step_count == 3 and fa.lr == 2e-4
    off = 0
    for i, p in enumerate(m1.parameters()):
        n = p.numel()
        assert torch.equal(fa.m[off:off + n], opt.state[p]["exp_avg"].flatten())
        assert torch.equal(fa.v[off:off + n], opt.state[p]["exp_avg_sq"].flatten())
        off += n
    # and back: a stock Adam accepts what FlatAdam writes
    opt2 = torch.optim.Adam(m2.parameters(), lr=1e-3)
    opt2.load_state_dict(fa.state_dict())
    assert opt2.param_groups[0]["lr"] == 2e-4
    for p1, p2 in zip(m1.parameters(), m2.parameters()):
        assert torch.equal(opt2.state[p2]["exp_avg"], opt.state[p1]["exp_avg"])
        assert float(opt2.state[p2]["step"]) == 3.0
    # a fresh FlatAdam has empty state, like a fresh torch Adam
    assert T.FlatAdam(flat, shapes=[p.shape for p in m2.parameters()]).state_dict()["state"] == {}


def test_flat_adam_state_resumes_in_the_reference_trainer():
    """The reference's resume (esrgan/train_esrgan.py:151-160): opt.load_state_dict(checkpoint state),
    then CosineAnnealingLR(opt, T_max, eta_min, last_epoch=start_epoch-1) — which raises KeyError
    unless the loaded parameter group carries 'initial_lr'."""
    T = pkg("train")
    F = pkg("utils.flat")
    m = nn.Sequential(nn.Linear(5, 4), nn.Linear(4, 3))
    flat = F.flatten_parameters(m)
    fa = T.FlatAdam(flat, lr=2e-4, shapes=[p.shape for p in m.parameters()])
    fa.lr = T.cosine_lr(2e-4, 7)                    # seven epochs into the schedule
    fa.step_count = 5
    fa.m.fill_(0.1)
    fa.v.fill_(0.2)
    sd = fa.state_dict()
    assert sd["param_groups"][0]["initial_lr"] == 2e-4
    start_epoch = 7
    opt = torch.optim.Adam(m.parameters(), lr=2e-4, betas=(0.9, 0.999))
    opt.load_state_dict(sd)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=1000, eta_min=1e-7, last_epoch=start_epoch - 1)
    assert sched.base_lrs == [2e-4]
    opt.step()
    sched.step()
    # the scheduler's constructor takes one step (to start_epoch) and the loop one more; torch's
    # recursive form continues from the loaded lr, so allow for its drift against the closed form
    assert abs(opt.param_groups[0]["lr"] - T.cosine_lr(2e-4, start_epoch + 1)) < 2e-8
    # and a state written by the reference's optimizer (scheduler attached) loads back
    fb = T.FlatAdam(flat, lr=1.0, shapes=[p.shape for p in m.parameters()])
    fb.load_state_dict(opt.state_dict())
    assert fb.initial_lr == 2e-4 and abs(fb.lr - T.cosine_lr(2e-4, start_epoch + 1)) < 2e-8


def test_training_checkpoint_round_trip(tmp_path):
    CK = pkg("utils.checkpointing")
    g, d = _small_models()
    og = torch.optim.Adam(g.parameters(), lr=2e-4)
    od = torch.optim.Adam(d.parameters(), lr=2e-4)
    for p in list(g.parameters()) + list(d.parameters()):
        p.grad = torch.full_like(p, 0.01)
    og.step()
    od.step()
    (tmp_path / "latest_checkpoint_epoch_3.pth").write_bytes(b"stale")
    path = CK.save_training_checkpoint(tmp_path, 4, g, d, og, od, best_psnr=21.5)
    assert path.name == "latest_checkpoint_epoch_4.pth"
    assert [f.name for f in tmp_path.glob("latest_checkpoint_*.pth")] == [path.name]
    raw = torch.load(path, map_location="cpu", weights_only=True)
    assert set(raw) == {"epoch", "gen_state_dict", "disc_state_dict", "opt_g_state_dict", "opt_d_state_dict",
                        "best_psnr"}
    assert list(raw["disc_state_dict"]) == list(d.state_dict())
    g2, d2 = _small_models()
    for p in list(g2.parameters()) + list(d2.parameters()):
        p.data.zero_()
    og2 = torch.optim.Adam(g2.parameters(), lr=1.0)
    od2 = torch.optim.Adam(d2.parameters(), lr=1.0)
    epoch, best = CK.load_latest_training_checkpoint(tmp_path, g2, d2, og2, od2)
    assert (epoch, best) == (4, 21.5)
    for (k, a), (_, b) in zip(g.state_dict().items(), g2.state_dict().items()):
        assert torch.equal(a, b), k
    for (k, a), (_, b) in zip(d.state_dict().items(), d2.state_dict().items()):
        assert torch.equal(a, b), k
    assert og2.param_groups[0]["lr"] == 2e-4
    assert CK.load_latest_training_checkpoint(tmp_path / "none", g2, d2, og2, od2) == (0, 0.0)


def test_generator_files_and_generic_schema(tmp_path):
    CK = pkg("utils.checkpointing")
    G = pkg("esrgan.generator")
    man, _ = golden()
    case = man["cases"]["checkpoint_g"]
    g = G.GeneratorESRGAN(num_features=case["num_features"], num_blocks=case["num_blocks"])
    CK.load_generator(g, os.path.join(ROOT, "tests", "golden", case["file"]))     # written by the reference
    CK.save_generator(g, tmp_path / "best_generator.pth")
    raw = torch.load(tmp_path / "best_generator.pth", weights_only=True)
    assert list(raw) == list(g.state_dict())
    opt = torch.optim.Adam(g.parameters(), lr=2e-4)
    CK.save_checkpoints(g, opt, 7, best_psnr=20.0, psnr=22.0, checkpoint_dir=tmp_path / "gen")
    assert sorted(f.name for f in (tmp_path / "gen").iterdir()) == ["best_model.pth", "latest_checkpoint_7.pth"]
    CK.save_checkpoints(g, opt, 8, best_psnr=22.0, psnr=21.0, checkpoint_dir=tmp_path / "gen")
    assert sorted(f.name for f in (tmp_path / "gen").iterdir()) == ["best_model.pth", "latest_checkpoint_8.pth"]
    g2 = G.GeneratorESRGAN(num_features=case["num_features"], num_blocks=case["num_blocks"])
    opt2 = torch.optim.Adam(g2.parameters(), lr=1.0)
    assert CK.load_latest_checkpoint(g2, opt2, tmp_path / "gen") == 9
    for (k, a), (_, b) in zip(g.state_dict().items(), g2.state_dict().items()):
        assert torch.equal(a, b), k
    # the wrapped layouts inference.py:135-138 accepts
    torch.save({"model_state_dict": g.state_dict()}, tmp_path / "w1.pth")
    torch.save({"gen_state_dict": g.state_dict(), "epoch": 1}, tmp_path / "w2.pth")
    for f in ("w1.pth", "w2.pth", "gen/latest_checkpoint_8.pth"):
        g3 = G.GeneratorESRGAN(num_features=case["num_features"], num_blocks=case["num_blocks"])
        CK.load_generator(g3, tmp_path / f)
        assert torch.equal(g3.conv_last.weight, g.conv_last.weight)


def test_cosine_lr_matches_torch_scheduler():
    T = pkg("train")
    p = nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=2e-4)
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=1000, eta_min=1e-7)
    for epoch in range(1, 6):
        opt.step()
        sch.step()
        assert abs(T.cosine_lr(2e-4, epoch) - opt.param_groups[0]["lr"]) < 1e-12
