"""Epoch-level training harness: the counterpart of the reference's
``esrgan/train_esrgan.py`` (``train_one_epoch`` :80-118, ``validate_one_epoch`` :43-77,
``main`` :121-208) on top of the libsrx trainers.

What is kept: the file names and layouts under the checkpoint directory
(``generator_pretrained.pth``, ``best_generator.pth``, ``latest_checkpoint_epoch_{n}.pth``
with the keys of :199-206), the resume rule (newest ``latest_checkpoint_*.pth``, :143-155),
``CosineAnnealingLR(T_max=epochs, eta_min=1e-7)`` stepped once per epoch for both optimizers
(:159-160,188), best-by-validation-PSNR (:191-194), the ``mode`` switch
``pretrain | train | all`` (:166,173,178).

What differs: the generator pre-training loop, an empty stub in the reference (:168-170), is
real here (L1 only, ``GeneratorWarmupTrainer``); bf16 needs no GradScaler; losses accumulate
on the device and are read once per epoch (the reference reads seven scalars per step,
:113-118); nothing is logged to wandb — ``log`` receives the same numbers.  Batches must
all have the shape the trainers were planned for (``batch_shape``): use a loader that
drops a ragged last batch.
"""
from __future__ import annotations

from pathlib import Path
from typing import Callable, Dict, Iterable, List, Optional

import torch

from .. import train as T
from ..parallel import barrier
from ..utils import checkpointing as ckpt
from ..validate import Validator

PRETRAINED_NAME = "generator_pretrained.pth"
BEST_NAME = "best_generator.pth"


def _check_shape(lr: torch.Tensor, batch_shape) -> None:
    if tuple(lr.shape) != tuple(batch_shape):
        raise ValueError(f"batch of shape {tuple(lr.shape)} but the trainer was planned for {tuple(batch_shape)}; "
                         f"drop the ragged last batch")


def train_one_epoch(trainer: T.AdversarialTrainer, loader: Iterable, batch_shape=None) -> Dict[str, float]:
    """One pass over the loader, one D step + one G step per batch (esrgan/train_esrgan.py:83-111).
    Returns the epoch means of the five losses the reference logs (:113-117)."""
    dev = trainer.device
    n = 0
    for lr, hr in loader:
        if batch_shape is not None:
            _check_shape(lr, batch_shape)
        trainer.step(lr.to(dev, non_blocking=True), hr.to(dev, non_blocking=True))
        n += 1
    if n == 0:
        raise ValueError("empty training loader")
    return trainer.losses()


def pretrain_one_epoch(trainer: T.GeneratorWarmupTrainer, loader: Iterable, batch_shape=None) -> float:
    """One L1-only pass over the loader (the loop the reference leaves out, :168-170).
    Returns the epoch's mean L1 loss."""
    dev = trainer.flat.device
    n = 0
    for lr, hr in loader:
        if batch_shape is not None:
            _check_shape(lr, batch_shape)
        trainer.step(lr.to(dev, non_blocking=True), hr.to(dev, non_blocking=True))
        n += 1
    if n == 0:
        raise ValueError("empty training loader")
    return trainer.mean_loss()


def validate_one_epoch(gen, loader: Iterable, validator: Optional[Validator] = None) -> float:
    """Average validation PSNR of the generator (what the reference returns, :77); the bicubic
    yardstick and the improvement are in ``Validator.validate_one_epoch``'s dictionary."""
    v = validator if validator is not None else Validator(gen)
    return v.validate_one_epoch(loader)["val_sr_psnr"]


def fit(gen, disc, vgg_loss, train_loader: Iterable, val_loader: Iterable, checkpoint_dir, *, batch_shape,
        epochs: int, pretrain_epochs: int = 0, mode: str = "all", resume: bool = False, lr: float = 2e-4,
        lambda_l1: float = 1.0, lambda_adv: float = 5e-3, lambda_percep: float = 1.0,
        compute_dtype=torch.bfloat16, grad_sync_g: Optional[Callable] = None,
        grad_sync_d: Optional[Callable] = None, log: Callable[[str], None] = print) -> List[Dict[str, float]]:
    """The body of the reference's ``main`` (:139-208) for modules that already sit on the GPU.
    ``train_loader`` / ``val_loader`` are re-iterated every epoch and yield ``(lr, hr)`` batches
    in [-1, 1]; ``batch_shape`` is the LR batch shape ``(B, 3, h, w)``.  Returns one dictionary
    per finished adversarial epoch (losses, learning rate, validation numbers)."""
    if mode not in ("pretrain", "train", "all"):
        raise ValueError(f"mode must be pretrain, train or all (got {mode!r})")
    cdir = Path(checkpoint_dir)
    cdir.mkdir(parents=True, exist_ok=True)
    pretrained = cdir / PRETRAINED_NAME

    adv = T.AdversarialTrainer(gen, disc, vgg_loss, batch_shape, compute_dtype, lr=lr, lambda_l1=lambda_l1,
                               lambda_adv=lambda_adv, lambda_percep=lambda_percep,
                               grad_sync_g=grad_sync_g, grad_sync_d=grad_sync_d)
    log(f"--- Perceptual loss: VGG19 weights from {getattr(vgg_loss, 'weights_source', 'unknown')} ---")
    start_epoch, best_psnr = 0, 0.0
    if resume:
        start_epoch, best_psnr = ckpt.load_latest_training_checkpoint(cdir, gen, disc, adv.opt_g, adv.opt_d)
        log(f"--- Resumed from epoch {start_epoch} (best PSNR = {best_psnr:.2f} dB) ---" if start_epoch
            else "--- No checkpoint found, starting fresh. ---")

    if mode in ("pretrain", "all") and not pretrained.exists() and start_epoch == 0 and pretrain_epochs > 0:
        log("--- Starting Generator Pre-training (L1 Loss only) ---")
        warm = T.GeneratorWarmupTrainer(gen, batch_shape, compute_dtype, lr=lr, grad_sync=grad_sync_g)
        for epoch in range(pretrain_epochs):
            l1 = pretrain_one_epoch(warm, train_loader, batch_shape)
            log(f"Pre-train epoch {epoch + 1} | L1: {l1:.5f}")
            ckpt.save_generator(gen, pretrained)
        del warm
        barrier()                     # rank 0's file is complete before any rank tests for it or reads it
        log(f"--- Pre-training Finished. Model saved to '{pretrained}' ---")

    barrier()
    if mode in ("train", "all") and start_epoch == 0 and pretrained.exists():
        log(f"--- Loading pre-trained generator from {pretrained} ---")
        ckpt.load_generator(gen, pretrained)

    history: List[Dict[str, float]] = []
    if mode not in ("train", "all"):
        return history
    validator = Validator(gen, compute_dtype)
    for epoch in range(start_epoch, epochs):
        cur_lr = T.cosine_lr(lr, epoch, t_max=epochs, eta_min=1e-7)
        adv.set_lr(cur_lr)
        rec = train_one_epoch(adv, train_loader, batch_shape)
        val = validator.validate_one_epoch(val_loader)
        avg_psnr = val["val_sr_psnr"]
        rec.update(val)
        rec.update(epoch=epoch + 1, lr=cur_lr)
        log(f"Epoch {epoch + 1} | Validation PSNR: {avg_psnr:.2f} dB | bicubic {val['val_bicubic_psnr']:.2f} dB | "
            f"loss_disc {rec['loss_disc']:.4f} loss_gen {rec['loss_gen']:.4f}")
        if avg_psnr > best_psnr:
            best_psnr = avg_psnr
            ckpt.save_generator(gen, cdir / BEST_NAME)
            log(f"New best generator saved with PSNR: {best_psnr:.2f} dB")
        rec["best_psnr"] = best_psnr
        ckpt.save_training_checkpoint(cdir, epoch + 1, gen, disc, adv.opt_g, adv.opt_d, best_psnr)
        history.append(rec)
    return history
