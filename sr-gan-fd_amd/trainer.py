"""Fused training iterations mirroring the reference's ``train()`` bodies, on the HIP engines.

  * ``GeneratorTrainer.step``  = ESRGAN/train_rrdbnet.py:244-272 / BSRGAN/train_bsrnet.py:244-272
    (zero_grad, G forward, L1 * weight, backward, Adam step, EMA update).
  * ``GanTrainer.step``        = BSRGAN/train_bsrgan.py:387-483 (see engine_d.py / gan.py).

They call the same engines as the ``nn.Module`` surface (``model.py``) but skip the autograd
bookkeeping: loss + dLoss/dSR come from one HIP kernel, the flat gradient goes straight to the fused
Adam + EMA kernel, and under data parallelism the flat gradient is all-reduced once (RCCL).
Mixed precision note: the reference runs ``amp.autocast()`` + ``GradScaler`` (fp16 on CUDA).  Here the
compute dtype is bf16 with fp32 master weights / fp32 accumulation, which needs no loss scaling, so
the scaler is the identity (as it is on the reference's CPU path).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from . import _abi as A
from .engine import generator_engine
from .parallel import allreduce_sum_


class FlatAdamEMA:
    """torch.optim.Adam maths (amsgrad=False) + AveragedModel(avg_fn=(1-d)*ema + d*p) over one flat buffer."""

    def __init__(self, flat: Tensor, lr: float, betas: Tuple[float, float], eps: float, weight_decay: float = 0.0,
                 ema_decay: Optional[float] = None):
        self.flat = flat
        self.m = torch.zeros_like(flat)
        self.v = torch.zeros_like(flat)
        self.ema = torch.zeros_like(flat) if ema_decay is not None else None
        self.lr, self.betas, self.eps, self.wd, self.ema_decay = lr, betas, eps, weight_decay, ema_decay
        self.t = 0
        self.n_averaged = 0
        self.step_dev: Optional[Tensor] = None     # device-side step counter (hipGraph replay), see graph.py
        self.bc_dev: Optional[Tensor] = None

    def use_device_step(self) -> None:
        """Keep the step count in device memory from now on (a captured graph cannot change kernel arguments)."""
        if self.step_dev is None:
            self.step_dev = torch.tensor([self.t], dtype=torch.int32, device=self.flat.device)
            self.bc_dev = torch.zeros(2, dtype=torch.float32, device=self.flat.device)

    def step(self, grad: Tensor, grad_scale: float = 1.0, update_ema: bool = True) -> None:
        self.t += 1
        mode = 0
        if self.ema is not None and update_ema:
            mode = 1 if self.n_averaged == 0 else 2
            self.n_averaged += 1
        if self.step_dev is not None:
            A.check(A.lib().srganfd_adam_ema_dev(self.flat.data_ptr(), grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                                 self.ema.data_ptr() if self.ema is not None else None, self.flat.numel(), self.lr,
                                                 self.betas[0], self.betas[1], self.eps, self.wd, self.step_dev.data_ptr(),
                                                 self.bc_dev.data_ptr(), grad_scale, self.ema_decay or 0.0, mode, A.stream_ptr()), "adam_ema_dev")
            return
        A.check(A.lib().srganfd_adam_ema(self.flat.data_ptr(), grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                         self.ema.data_ptr() if self.ema is not None else None, self.flat.numel(), self.lr,
                                         self.betas[0], self.betas[1], self.eps, self.wd, self.t, grad_scale,
                                         self.ema_decay or 0.0, mode, A.stream_ptr()), "adam_ema")


class GeneratorTrainer:
    """Generator-only iteration (BASELINE.json configs[0] and [1])."""

    def __init__(self, g_model, lr: float, betas=(0.9, 0.99), eps: float = 1e-8, weight_decay: float = 0.0,
                 ema_decay: Optional[float] = 0.999, loss_weight: float = 1.0, process_group=None):
        self.g = g_model
        self.eng = generator_engine(g_model)
        dev = next(g_model.parameters()).device
        self.flat = self.eng.fp.sync(dev)
        self.opt = FlatAdamEMA(self.flat, lr, betas, eps, weight_decay, ema_decay)
        self.loss_weight = loss_weight
        self.pg = process_group
        self.loss_buf = torch.zeros(1, dtype=torch.float32, device=dev)
        self.ws = torch.empty(A.LOSS_WS_FLOATS, dtype=torch.float32, device=dev)
        self.dsr: Optional[Tensor] = None

    def step(self, lr_img: Tensor, gt: Tensor) -> Tensor:
        """Returns the (device, 1-element) loss tensor; no host sync inside."""
        eng = self.eng
        sr = eng.forward(lr_img, True)
        sp, token = eng._last, eng.token
        if self.dsr is None or self.dsr.shape != sr.shape:
            self.dsr = torch.empty_like(sr)
        gt = gt.contiguous().float()
        A.check(A.lib().srganfd_l1_loss(sr.data_ptr(), gt.data_ptr(), sr.numel(), self.loss_weight, self.loss_buf.data_ptr(), 0,
                                        self.dsr.data_ptr(), self.loss_weight, self.ws.data_ptr(), A.stream_ptr()), "l1_loss")
        grad, _ = eng.backward(sp, token, self.dsr, False)
        scale = allreduce_sum_(grad, self.pg)          # RCCL over xGMI: ONE flat 67 MB buffer per step
        self.opt.step(grad, scale)
        eng.fp._seen = None                             # parameters changed behind autograd's back -> re-pack
        self.sr = sr
        return self.loss_buf
