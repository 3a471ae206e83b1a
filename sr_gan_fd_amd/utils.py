"""Checkpoints: the reference's files load into the drop-in modules and into the fused trainers, and the fused trainers write
files the reference's scripts can resume from.

``load_state_dict`` keeps the call and return conventions of BSRGAN/utils.py:34-81 (default mode: weights only, entries whose key
or shape does not match are dropped silently; ``"resume"``: epoch, best metrics, optimizer, scheduler, EMA as well).  Besides
torch optimizers / ``AveragedModel`` it accepts the fused path's objects: a ``trainer.FlatAdamEMA`` as ``optimizer`` (per-parameter
``exp_avg`` / ``exp_avg_sq`` / ``step`` scattered into its flat moments) and the same object as ``ema_model`` (``module.<name>``
entries + ``n_averaged`` into its flat EMA copy).  ``save_checkpoint`` writes the dictionary train_bsrnet.py:124-130 writes."""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn


def _matching(src: dict, dst: dict, check_shape: bool) -> dict:
    return {k: v for k, v in src.items() if k in dst and (not check_shape or v.size() == dst[k].size())}


def load_state_dict(model: nn.Module, model_weights_path: str, ema_model=None, optimizer=None, scheduler=None,
                    load_mode: Optional[str] = None):
    ckpt = torch.load(model_weights_path, map_location="cpu")
    resume = load_mode == "resume"
    merged = model.state_dict()
    merged.update(_matching(ckpt["state_dict"], merged, check_shape=not resume))     # utils.py:51-53 (resume) / :74-77 (default)
    model.load_state_dict(merged)
    if not resume:
        return model
    optimizer.load_state_dict(ckpt["optimizer"])                # torch.optim.Adam or trainer.FlatAdamEMA: same format
    if scheduler is not None:
        scheduler.load_state_dict(ckpt["scheduler"])
    if ema_model is not None:
        if hasattr(ema_model, "load_ema_state_dict"):           # fused path: the optimizer object owns the EMA copy
            ema_model.load_ema_state_dict(ckpt["ema_state_dict"])
        else:
            ema_sd = ema_model.state_dict()
            ema_sd.update(_matching(ckpt["ema_state_dict"], ema_sd, check_shape=False))
            ema_model.load_state_dict(ema_sd)
    return model, ema_model, ckpt["epoch"], ckpt["best_psnr"], ckpt["best_ssim"], optimizer, scheduler


def save_checkpoint(path: str, model: nn.Module, optimizer, ema=None, scheduler=None, epoch: int = 0, best_psnr: float = 0.0,
                    best_ssim: float = 0.0) -> None:
    """The dictionary of train_bsrnet.py:124-130 / train_bsrgan.py:203-260.  ``optimizer``: torch optimizer or FlatAdamEMA;
    ``ema``: ``AveragedModel`` or the FlatAdamEMA that owns the EMA copy."""
    out = {"epoch": epoch, "best_psnr": best_psnr, "best_ssim": best_ssim, "state_dict": model.state_dict(), "optimizer": optimizer.state_dict()}
    if ema is not None:
        out["ema_state_dict"] = ema.ema_state_dict() if hasattr(ema, "ema_state_dict") else ema.state_dict()
    if scheduler is not None:
        out["scheduler"] = scheduler.state_dict()
    torch.save(out, path)
