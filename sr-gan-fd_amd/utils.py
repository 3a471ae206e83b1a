"""Checkpoint loading with the reference's semantics (BSRGAN/utils.py:34-81), so that upstream files such as
``BSRGAN_x4-DIV2K-6d507222.pth.tar`` and the train scripts' own ``g_last.pth.tar`` load into the drop-in modules."""
from __future__ import annotations

import torch
from torch import nn


def load_state_dict(model: nn.Module, model_weights_path: str, ema_model: nn.Module = None, optimizer=None, scheduler=None,
                    load_mode: str = None):
    """Same call and return conventions as the reference: default mode copies the checkpoint's ``state_dict`` entries
    whose key exists in the model AND whose shape matches, silently dropping the rest (utils.py:73-79); ``"resume"``
    also restores epoch / best metrics / optimizer / scheduler / EMA state (utils.py:45-69)."""
    checkpoint = torch.load(model_weights_path, map_location=lambda storage, loc: storage)
    if load_mode == "resume":
        start_epoch, best_psnr, best_ssim = checkpoint["epoch"], checkpoint["best_psnr"], checkpoint["best_ssim"]
        model_state_dict = model.state_dict()
        model_state_dict.update({k: v for k, v in checkpoint["state_dict"].items() if k in model_state_dict.keys()})
        model.load_state_dict(model_state_dict)
        optimizer.load_state_dict(checkpoint["optimizer"])
        if scheduler is not None:
            scheduler.load_state_dict(checkpoint["scheduler"])
        if ema_model is not None:
            ema_model_state_dict = ema_model.state_dict()
            ema_model_state_dict.update({k: v for k, v in checkpoint["ema_state_dict"].items() if k in ema_model_state_dict.keys()})
            ema_model.load_state_dict(ema_model_state_dict)
        return model, ema_model, start_epoch, best_psnr, best_ssim, optimizer, scheduler
    model_state_dict = model.state_dict()
    model_state_dict.update({k: v for k, v in checkpoint["state_dict"].items()
                             if k in model_state_dict.keys() and v.size() == model_state_dict[k].size()})
    model.load_state_dict(model_state_dict)
    return model
