"""Checkpoint files in the reference's three layouts, readable and writable both ways:

* raw generator ``state_dict`` — ``best_generator.pth`` / ``generator_pretrained.pth``
  (esrgan/train_esrgan.py:170,193; what ``inference.py:133-141`` loads);
* the trainer's resume file ``latest_checkpoint_epoch_{N}.pth`` =
  ``{epoch, gen_state_dict, disc_state_dict, opt_g_state_dict, opt_d_state_dict, best_psnr}``
  (esrgan/train_esrgan.py:197-207), resumed from the newest file by ctime (:143-155);
* the generic helper schema ``latest_checkpoint_{epoch}.pth`` = ``{model, optimizer, epoch,
  psnr}`` + ``best_model.pth`` (utils/checkpointing.py:23-44), whose two functions keep their
  names and argument order here.

Optimizers may be ``torch.optim.Adam`` or this package's ``FlatAdam`` (its state_dict is in
torch's layout, so files are interchangeable).  Files are read with ``weights_only=True``.
In a data-parallel job only rank 0 writes; every rank can read.
"""
from __future__ import annotations

import os
import re
from pathlib import Path
from typing import Optional, Tuple

import torch


def _is_writer() -> bool:
    import torch.distributed as dist
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0


def _load(path):
    return torch.load(os.fspath(path), map_location="cpu", weights_only=True)


def _cpu_state(sd):
    """Detached CPU copy that keeps the ``_metadata`` attribute (spectral-norm version tags)."""
    out = type(sd)((k, v.detach().cpu().clone() if torch.is_tensor(v) else v) for k, v in sd.items())
    if hasattr(sd, "_metadata"):
        out._metadata = sd._metadata
    return out


# ------------------------------------------------------------------ raw generator files
def save_generator(gen, path) -> None:
    """``torch.save(gen.state_dict(), path)`` (esrgan/train_esrgan.py:170,193), rank 0 only."""
    if _is_writer():
        Path(path).parent.mkdir(parents=True, exist_ok=True)
        torch.save(_cpu_state(gen.state_dict()), os.fspath(path))


def load_generator(gen, path, strict: bool = True) -> None:
    """Raw state_dict, or either wrapped layout (inference.py:135-138)."""
    sd = _load(path)
    if isinstance(sd, dict) and "model_state_dict" in sd:
        sd = sd["model_state_dict"]
    elif isinstance(sd, dict) and "gen_state_dict" in sd:
        sd = sd["gen_state_dict"]
    elif isinstance(sd, dict) and "model" in sd and isinstance(sd["model"], dict):
        sd = sd["model"]
    gen.load_state_dict(sd, strict=strict)


# --------------------------------------------------------------- trainer resume files
def save_training_checkpoint(checkpoint_dir, epoch: int, gen, disc, opt_g, opt_d, best_psnr: float) -> Optional[Path]:
    """esrgan/train_esrgan.py:196-207: drop older ``latest_checkpoint_*.pth``, write
    ``latest_checkpoint_epoch_{epoch}.pth``.  ``epoch`` is the number of finished epochs."""
    if not _is_writer():
        return None
    d = Path(checkpoint_dir)
    d.mkdir(parents=True, exist_ok=True)
    for old in d.glob("latest_checkpoint_*.pth"):
        old.unlink()
    path = d / f"latest_checkpoint_epoch_{epoch}.pth"
    torch.save({"epoch": epoch,
                "gen_state_dict": _cpu_state(gen.state_dict()),
                "disc_state_dict": _cpu_state(disc.state_dict()),
                "opt_g_state_dict": opt_g.state_dict(),
                "opt_d_state_dict": opt_d.state_dict(),
                "best_psnr": best_psnr}, os.fspath(path))
    return path


def load_latest_training_checkpoint(checkpoint_dir, gen, disc, opt_g, opt_d) -> Tuple[int, float]:
    """esrgan/train_esrgan.py:143-155.  Returns (start_epoch, best_psnr); (0, 0.0) if the
    directory holds no resume file."""
    files = list(Path(checkpoint_dir).glob("latest_checkpoint_*.pth"))
    if not files:
        return 0, 0.0
    ck = _load(max(files, key=os.path.getctime))
    gen.load_state_dict(ck["gen_state_dict"])
    disc.load_state_dict(ck["disc_state_dict"])
    opt_g.load_state_dict(ck["opt_g_state_dict"])
    opt_d.load_state_dict(ck["opt_d_state_dict"])
    return int(ck["epoch"]), float(ck.get("best_psnr", 0.0))


# --------------------------------------------- generic helper schema (utils/checkpointing.py)
def load_latest_checkpoint(model, optimizer, checkpoint_dir):
    """utils/checkpointing.py:7-21: newest ``latest_checkpoint_{epoch}.pth`` by epoch number;
    returns the epoch to start from (stored epoch + 1, or 0)."""
    ckpt_dir = Path(checkpoint_dir)
    ckpt_dir.mkdir(exist_ok=True, parents=True)
    start_epoch = 0
    ckpts = [f for f in ckpt_dir.glob("latest_checkpoint_*.pth") if re.search(r"latest_checkpoint_(\d+).pth", f.name)]
    if ckpts:
        latest = max(ckpts, key=lambda f: int(re.search(r"latest_checkpoint_(\d+).pth", f.name).group(1)))
        ck = _load(latest)
        model.load_state_dict(ck["model"])
        optimizer.load_state_dict(ck["optimizer"])
        start_epoch = ck.get("epoch", 0) + 1
    return start_epoch


def save_checkpoints(model, optimizer, epoch, best_psnr, psnr, checkpoint_dir):
    """utils/checkpointing.py:23-44."""
    if not _is_writer():
        return
    ckpt_dir = Path(checkpoint_dir)
    ckpt_dir.mkdir(exist_ok=True, parents=True)
    for f in ckpt_dir.glob("latest_checkpoint_*.pth"):
        os.remove(f)
    torch.save({"model": _cpu_state(model.state_dict()), "optimizer": optimizer.state_dict(),
                "epoch": epoch, "psnr": psnr}, os.fspath(ckpt_dir / f"latest_checkpoint_{epoch}.pth"))
    if psnr > best_psnr:
        torch.save(_cpu_state(model.state_dict()), os.fspath(ckpt_dir / "best_model.pth"))
