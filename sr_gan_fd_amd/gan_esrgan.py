"""Fused relativistic-average GAN iteration of ESRGAN -- ESRGAN/train_esrgan.py:340-431 on the HIP engines.

Order and arithmetic of the reference's loop body:
  generator (discriminator frozen, :361-392): sr = G(lr); gt_out = D(gt); sr_out = D(sr); pixel = w_p * L1(sr, gt);
    content = w_c * L1(vgg(sr)[node], vgg(gt)[node]) (differentiable, one node); adversarial = w_a * 0.5 * (BCE(gt_out - mean(sr_out), 0)
    + BCE(sr_out - mean(gt_out), 1)); backward of the sum into the generator only -- through D's data gradient with BOTH halves of the
    adversarial term reaching sr_out (the first through the mean) -- Adam step, EMA update;
  discriminator (:394-425): gt_out = D(gt); sr_out = D(sr.detach()); 0.5 * BCE(gt_out - mean(sr_out), 1) backward (retain_graph);
    sr_out = D(sr.detach()) again; 0.5 * BCE(sr_out - mean(gt_out), 0) backward; Adam step.
Five training-mode discriminator forwards per iteration (BatchNorm running statistics advance five times, as in the reference).
The two discriminator losses are back-propagated as TWO passes instead of the reference's four graph traversals: the second and third
forwards of sr.detach() are the same function of the same inputs, so the gradients that reach D through them add up to one pass through
the last one with the summed seeds (d loss_gt / d mean(sr_out) + d loss_sr / d sr_out), and likewise for the gt forward.
Every loss, seed, reduction and optimizer step is a HIP kernel (srganfd_l1_loss, srganfd_bce_logits_relativistic, the engines'
convolutions, srganfd_adam_ema); nothing runs as an ATen op.  Mixed precision: trainer.LossScaler, one instance for both networks (:103).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import Tensor

from . import _abi as A
from .engine import _engine, generator_engine
from .engine_e import esrgan_discriminator_engine
from .engine_v import ContentLossGradEngine
from .parallel import BucketReducer, allreduce_sum_
from .trainer import FlatAdamEMA, GanCheckpointMixin, LossScaler, check_loss_scaling, needs_loss_scaling, pin_training_dtype


class EsrganGanTrainer(GanCheckpointMixin):
    def __init__(self, g_model, d_model, content_criterion=None, *, g_lr: float = 1e-4, d_lr: float = 1e-4, betas=(0.9, 0.99),
                 eps: float = 1e-8, weight_decay: float = 0.0, ema_decay: Optional[float] = 0.99998, pixel_weight: float = 0.01,
                 content_weight: float = 1.0, adversarial_weight: float = 0.005, process_group=None):
        # defaults = ESRGAN/esrgan_config.py:75-111
        self.g, self.d, self.content = g_model, d_model, content_criterion
        pin_training_dtype(g_model, d_model, content_criterion)     # train_esrgan.py:370-410 runs under amp.autocast(): float16
        self.ge, self.de = generator_engine(g_model), esrgan_discriminator_engine(d_model)
        if content_criterion is not None and not getattr(content_criterion, "single_node", False):
            raise A.SrganfdError("EsrganGanTrainer: content_criterion must be model.ContentLoss built with ONE node name, a str (ESRGAN's form)")
        self.ce = _engine(content_criterion, lambda: ContentLossGradEngine(content_criterion)) if content_criterion is not None else None
        dev = next(g_model.parameters()).device
        self.dev = dev
        self.g_opt = FlatAdamEMA(self.ge.fp.sync(dev), g_lr, betas, eps, weight_decay, ema_decay, layout=self.ge.fp)
        self.d_opt = FlatAdamEMA(self.de.fp.sync(dev), d_lr, betas, eps, weight_decay, None, layout=self.de.fp)
        self.pw, self.cw, self.aw = float(pixel_weight), float(content_weight), float(adversarial_weight)
        self.scaler = LossScaler(dev, enabled=needs_loss_scaling(g_model, d_model, content_criterion))
        # Under data parallelism the discriminator's BatchNorm layers (ESRGAN/model.py:98-126) use per-rank batch statistics, which is
        # what the unconverted reference does under DistributedDataParallel; there is no SyncBatchNorm hook for this discriminator.
        self.pg = process_group
        self.g_reducer = BucketReducer(dev, process_group)
        # [d_loss, pixel, content, adversarial, D(gt) prob, D(sr) prob, 0, 0]  (train_esrgan.py:416,430-431)
        self.scalars = torch.zeros(8, dtype=torch.float32, device=dev)
        self.ws = torch.empty(A.LOSS_WS_FLOATS, dtype=torch.float32, device=dev)
        self._bufs: Dict[tuple, Tensor] = {}
        self.sr: Optional[Tensor] = None

    def _buf(self, name, like: Tensor) -> Tensor:
        b = self._bufs.get((name, tuple(like.shape)))
        if b is None:
            b = torch.empty_like(like)
            self._bufs[(name, tuple(like.shape))] = b
        return b

    def _rel(self, x: Tensor, other: Tensor, target: float, weight: float, slot: int, accumulate: int, grad_x, acc_x: int, grad_other, acc_o: int,
             seed_weight: Optional[float] = None) -> None:
        """weight * mean BCE(x - mean(other), target) into scalars[slot]; seeds scaled by seed_weight (default: weight) x loss scale"""
        A.check(A.lib().srganfd_bce_logits_relativistic(
            x.data_ptr(), x.numel(), other.data_ptr(), other.numel(), target, weight, self.scalars.data_ptr() + 4 * slot, accumulate,
            grad_x.data_ptr() if grad_x is not None else None, acc_x, grad_other.data_ptr() if grad_other is not None else None, acc_o,
            weight if seed_weight is None else seed_weight, self.scaler.seed_ptr, self.ws.data_ptr(), A.stream_ptr()), "bce_logits_relativistic")

    def step(self, lr_img: Tensor, gt: Tensor) -> Tensor:
        """One iteration; returns the device tensor [d_loss, pixel, content, adversarial, D(gt), D(sr), 0, 0] (no host synchronisation)."""
        L, st = A.lib(), A.stream_ptr()
        ge, de = self.ge, self.de
        gt = gt.contiguous().float()
        check_loss_scaling(self.scaler, self.g, self.d, self.content)
        s = self.scalars.data_ptr()
        # ---- generator (train_esrgan.py:361-392) ----
        sr = ge.forward(lr_img, True)
        g_sp, g_tok = ge._last, ge.token
        gt_out = de.forward(gt, True)                                        # constant for the generator: only its mean is used
        sr_out = de.forward(sr, True)
        d_sp, d_tok = de._last, de.token
        dsr = self._buf("dsr", sr)
        A.check(L.srganfd_l1_loss(sr.data_ptr(), gt.data_ptr(), sr.numel(), self.pw, s + 4, 0, dsr.data_ptr(), self.pw, self.scaler.seed_ptr,
                                  self.ws.data_ptr(), st), "l1_loss")
        dl = self._buf("dl", sr_out)
        half = 0.5 * self.aw
        self._rel(gt_out, sr_out, 0.0, half, 3, 0, None, 0, dl, 0)           # reaches sr_out through the mean
        self._rel(sr_out, gt_out, 1.0, half, 3, 1, dl, 1, None, 0)           # reaches sr_out directly; loss values add up in slot 3
        _, dsr_adv = de.backward(d_sp, d_tok, dl, False, True)               # frozen D: data gradient only
        A.check(L.srganfd_axpby(A.View(dsr_adv.data_ptr(), 1, 0), A.View(dsr.data_ptr(), 1, 0), A.F32, dsr.numel(), 1, 1.0, 1.0, st), "axpby")
        if self.ce is not None:
            closs = self.ce.forward(sr, gt)                                  # 1-element device tensor: L1 at the node
            c_sp, c_tok = self.ce._last, self.ce.token
            A.check(L.srganfd_axpby(A.View(closs.data_ptr(), 1, 0), A.View(s + 8, 1, 0), A.F32, 1, 1, self.cw, 0.0, st), "axpby")
            dsr_c = self.ce.backward(c_sp, c_tok, None, self.cw, self.scaler.seed_ptr)
            A.check(L.srganfd_axpby(A.View(dsr_c.data_ptr(), 1, 0), A.View(dsr.data_ptr(), 1, 0), A.F32, dsr.numel(), 1, 1.0, 1.0, st), "axpby")
        self.g_reducer.begin()
        gg, _ = ge.backward(g_sp, g_tok, dsr, False, on_ready=self.g_reducer.bucket)
        self.scaler.step(self.g_opt, gg, self.g_reducer.finish())           # scaler.step(g_optimizer); scaler.update(); EMA (:383-389)
        ge.fp.touch()
        # ---- discriminator (train_esrgan.py:394-425) ----
        gt_out = de.forward(gt, True)
        gt_sp, gt_tok = de._last, de.token
        de.forward(sr, True)                                                 # second forward of the pair: advances the BatchNorm statistics
        sr_out = de.forward(sr, True)                                        # the same logits; this graph takes both seeds
        sr_sp, sr_tok = de._last, de.token
        dgt, dsr_l = self._buf("dgt", gt_out), self._buf("dl", sr_out)
        self._rel(gt_out, sr_out, 1.0, 0.5, 0, 0, dgt, 0, dsr_l, 0)          # d_loss_gt: seeds both graphs
        self._rel(sr_out, gt_out, 0.0, 0.5, 0, 1, dsr_l, 1, dgt, 1)          # d_loss_sr; d_loss = their sum in slot 0
        gd1, _ = de.backward(gt_sp, gt_tok, dgt, True, False)
        gd2, _ = de.backward(sr_sp, sr_tok, dsr_l, True, False)
        A.check(L.srganfd_axpby(A.View(gd1.data_ptr(), 1, 0), A.View(gd2.data_ptr(), 1, 0), A.F32, gd2.numel(), 1, 1.0, 1.0, st), "axpby")
        self.scaler.step(self.d_opt, gd2, allreduce_sum_(gd2, self.pg))      # scaler.step(d_optimizer); scaler.update() (:419-420)
        de.fp.touch()
        A.check(L.srganfd_sigmoid_of_mean(gt_out.data_ptr(), gt_out.numel(), s + 16, self.ws.data_ptr(), st), "sigmoid_of_mean")
        A.check(L.srganfd_sigmoid_of_mean(sr_out.data_ptr(), sr_out.numel(), s + 20, self.ws.data_ptr(), st), "sigmoid_of_mean")
        self.sr = sr
        return self.scalars
