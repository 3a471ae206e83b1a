"""Fused GAN training iteration -- the reference's BSRGAN/train_bsrgan.py:387-483 on the HIP engines
(``generator_first=True``: Real_ESRGAN/train_realesrgan.py:407-476, see ``GanTrainer._step_generator_first``).

Order kept exactly: D(gt) forward+backward, G forward, D(sr.detach())
forward+backward (gradients accumulate), D Adam step, freeze D, pixel (L1) + content (VGG-19, detached:
logged only) + adversarial (BCE vs ones through the UPDATED D; spectral-norm u/v advance a third time),
G backward (through D's data gradient only), G Adam step, EMA update.
Data parallel: one flat-gradient all-reduce per network per iteration (D after its second backward,
G after its backward) over RCCL; losses are means over the local shard, identical maths to the
single-process step on the concatenated batch.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import Tensor

from . import _abi as A
from .engine import generator_engine
from .engine_a import aesrgan_engine
from .engine_d import discriminator_engine
from .parallel import BucketReducer, SideStreamReducer, SyncBatchNormReduce, allreduce_sum_
from .trainer import FlatAdamEMA, GanCheckpointMixin, LossScaler, check_loss_scaling, needs_loss_scaling, pin_training_dtype


class GanTrainer(GanCheckpointMixin):
    def __init__(self, g_model, d_model, content_criterion=None, *, g_lr: float = 8e-5, d_lr: float = 2e-4, betas=(0.9, 0.999),
                 eps: float = 1e-4, weight_decay: float = 0.0, ema_decay: float = 0.999, pixel_weight: float = 20.0,
                 content_weight=1.0, adversarial_weight: float = 0.5, train_generator: bool = True, process_group=None,
                 generator_first: bool = False, sync_batchnorm: bool = False):
        # defaults = BSRGAN/bsrgan_config.py:137-159
        self.g, self.d, self.content = g_model, d_model, content_criterion
        pin_training_dtype(g_model, d_model, content_criterion)     # modules left at "follow autocast" train in the loops' float16
        # either discriminator of the reference: DiscriminatorUNet (BSRGAN / Real-ESRGAN) or the A-ESRGAN attention U-Net
        # (A-ESRGAN/train_aesrgan.py:396-483 runs the same statements around it); both engines share one interface
        d_engine = aesrgan_engine if type(d_model).__name__ == "UNetDiscriminatorAesrgan" else discriminator_engine
        self.ge, self.de = generator_engine(g_model), d_engine(d_model)
        dev = next(g_model.parameters()).device
        self.dev = dev
        self.g_opt = FlatAdamEMA(self.ge.fp.sync(dev), g_lr, betas, eps, weight_decay, ema_decay, layout=self.ge.fp)
        self.d_opt = FlatAdamEMA(self.de.fp.sync(dev), d_lr, betas, eps, weight_decay, None, layout=self.de.fp)
        self.pw, self.cw, self.aw = pixel_weight, content_weight, adversarial_weight
        self.scaler = LossScaler(dev, enabled=needs_loss_scaling(g_model, d_model))   # ONE scaler for both networks (train_bsrgan.py:109)
        # per-node content weights (Real-ESRGAN's list): built once -- a host list turned into a device tensor inside step() is a
        # host->device copy per iteration and cannot be captured into a graph
        self._cw_t = None if isinstance(content_weight, (int, float)) else torch.tensor(list(content_weight), dtype=torch.float32, device=dev)
        self.train_generator = train_generator
        self.generator_first = generator_first
        self.pg = process_group
        self.d_reducer = SideStreamReducer(dev, process_group)      # D's gradient exchange + Adam beside the generator-side losses
        self.g_reducer = BucketReducer(dev, process_group)          # G's gradient exchange in buckets behind its own backward pass
        if sync_batchnorm:
            # A-ESRGAN's attention blocks carry BatchNorm2d (A-ESRGAN/model.py:233): whole-batch statistics across the ranks
            if not hasattr(self.de, "sync_bn"):
                raise ValueError("sync_batchnorm: this discriminator has no BatchNorm layers served by the two-phase kernels")
            self.de.sync_bn = SyncBatchNormReduce(process_group)
        # [d_loss_hr, d_loss_sr, pixel, adversarial, D(gt) prob, D(sr) prob]
        self.scalars = torch.zeros(8, dtype=torch.float32, device=dev)
        self.ws = torch.empty(A.LOSS_WS_FLOATS, dtype=torch.float32, device=dev)
        self.content_vals: Optional[Tensor] = None
        self._bufs: Dict[tuple, Tensor] = {}

    def _buf(self, name, like: Tensor) -> Tensor:
        b = self._bufs.get((name, tuple(like.shape)))
        if b is None:
            b = torch.empty_like(like)
            self._bufs[(name, tuple(like.shape))] = b
        return b

    def _allreduce(self, grad: Tensor) -> float:
        return allreduce_sum_(grad, self.pg)

    def _bce(self, logits: Tensor, target: float, weight: float, slot: int, prob_slot: Optional[int], dlogits: Tensor) -> None:
        """loss value weighted by ``weight``; its gradient by ``weight`` times the loss scale the device holds when the kernel runs
        (scaler.scale(loss): trainer.LossScaler)"""
        s = self.scalars.data_ptr()
        A.check(A.lib().srganfd_bce_logits(logits.data_ptr(), logits.numel(), target, weight, s + 4 * slot, 0,
                                           (s + 4 * prob_slot) if prob_slot is not None else None, dlogits.data_ptr(), weight,
                                           self.scaler.seed_ptr, self.ws.data_ptr(), A.stream_ptr()), "bce_logits")

    def _content(self, sr: Tensor, gt: Tensor) -> None:
        if self.content is not None:
            cw = self.cw if self._cw_t is None else self._cw_t
            self.content_vals = self.content(sr, gt) * cw      # per-node weights broadcast over the (1, nodes) tensor

    def _step_generator_first(self, lr_img: Tensor, gt: Tensor, gt_usm: Optional[Tensor]) -> Tensor:
        """Real_ESRGAN/train_realesrgan.py:407-476: the GENERATOR step first -- pixel and (detached) content loss against the
        USM-sharpened GT, adversarial BCE vs ones through the frozen current D, G Adam step + EMA -- then the discriminator on the
        plain GT and on the detached SR (two backward passes accumulate), D Adam step.  Slots 4 / 5 hold sigmoid(mean(logits))
        as that script logs them (:475-476)."""
        L, st = A.lib(), A.stream_ptr()
        ge, de = self.ge, self.de
        gt = gt.contiguous().float()
        gtu = gt if gt_usm is None else gt_usm.contiguous().float()
        sr = ge.forward(lr_img, True)
        g_sp, g_tok = ge._last, ge.token
        dsr = self._buf("dsr", sr)
        check_loss_scaling(self.scaler, self.g, self.d)
        A.check(L.srganfd_l1_loss(sr.data_ptr(), gtu.data_ptr(), sr.numel(), self.pw, self.scalars.data_ptr() + 8, 0, dsr.data_ptr(), self.pw,
                                  self.scaler.seed_ptr, self.ws.data_ptr(), st), "l1_loss")
        self._content(sr, gtu)
        adv_out = de.forward(sr, True)
        dl = self._buf("dl", adv_out)
        self._bce(adv_out, 1.0, self.aw, 3, None, dl)
        if self.train_generator:
            _, dsr_adv = de.backward(de._last, de.token, dl, False, True)
            A.check(L.srganfd_axpby(A.View(dsr_adv.data_ptr(), 1, 0), A.View(dsr.data_ptr(), 1, 0), A.F32, dsr.numel(), 1, 1.0, 1.0, st), "axpby")
            self.g_reducer.begin()
            gg, _ = ge.backward(g_sp, g_tok, dsr, False, on_ready=self.g_reducer.bucket)
            self.scaler.step(self.g_opt, gg, self.g_reducer.finish())
            ge.fp.touch()
        s = self.scalars.data_ptr()
        gt_out = de.forward(gt, True)
        self._bce(gt_out, 1.0, 1.0, 0, None, dl)
        A.check(L.srganfd_sigmoid_of_mean(gt_out.data_ptr(), gt_out.numel(), s + 16, self.ws.data_ptr(), st), "sigmoid_of_mean")
        gd1, _ = de.backward(de._last, de.token, dl, True, False)
        sr_out = de.forward(sr, True)
        self._bce(sr_out, 0.0, 1.0, 1, None, dl)
        A.check(L.srganfd_sigmoid_of_mean(sr_out.data_ptr(), sr_out.numel(), s + 20, self.ws.data_ptr(), st), "sigmoid_of_mean")
        gd2, _ = de.backward(de._last, de.token, dl, True, False)
        A.check(L.srganfd_axpby(A.View(gd1.data_ptr(), 1, 0), A.View(gd2.data_ptr(), 1, 0), A.F32, gd2.numel(), 1, 1.0, 1.0, st), "axpby")
        self.scaler.step(self.d_opt, gd2, self._allreduce(gd2))
        self.sr = sr
        return self.scalars

    def step(self, lr_img: Tensor, gt: Tensor, gt_usm: Optional[Tensor] = None) -> Tensor:
        """One iteration; returns the device tensor [d_loss_hr, d_loss_sr, pixel, adversarial, D(gt), D(sr), 0, 0]
        (no host synchronisation inside; content-loss values are in ``self.content_vals``).  ``gt_usm`` (generator-first
        mode): the sharpened GT the generator's pixel / content losses compare against; the discriminator sees ``gt``."""
        if self.generator_first:
            return self._step_generator_first(lr_img, gt, gt_usm)
        if gt_usm is not None:
            raise A.SrganfdError("GanTrainer.step: gt_usm belongs to the generator-first (Real-ESRGAN) iteration")
        L, st = A.lib(), A.stream_ptr()
        ge, de = self.ge, self.de
        gt = gt.contiguous().float()
        # ---- discriminator ----
        check_loss_scaling(self.scaler, self.g, self.d)
        gt_out = de.forward(gt, True)
        dl = self._buf("dl", gt_out)
        self._bce(gt_out, 1.0, 1.0, 0, 4, dl)                 # scaler.scale(d_loss_*): train_bsrgan.py:420,430
        gd1, _ = de.backward(de._last, de.token, dl, True, False)
        sr = ge.forward(lr_img, True)
        g_sp, g_tok = ge._last, ge.token
        sr_out = de.forward(sr, True)
        self._bce(sr_out, 0.0, 1.0, 1, 5, dl)
        gd2, _ = de.backward(de._last, de.token, dl, True, False)
        A.check(L.srganfd_axpby(A.View(gd1.data_ptr(), 1, 0), A.View(gd2.data_ptr(), 1, 0), A.F32, gd2.numel(), 1, 1.0, 1.0, st), "axpby")
        # scaler.step(d_optimizer); scaler.update()  (:436-437).  Under data parallelism the all-reduce and the Adam kernel go to a
        # side stream: the pixel loss and the VGG-19 content forwards below do not read D's parameters
        self.d_reducer.launch(lambda: self.scaler.step(self.d_opt, gd2, self._allreduce(gd2)), tensors=(gd2,))
        # ---- generator ----
        dsr = self._buf("dsr", sr)
        self._content(sr, gt)
        self.d_reducer.wait()                               # D's updated parameters -- and the loss scale after its update() -- from here on
        # the pixel loss seeds the generator's backward pass with the scale AFTER the discriminator's scaler.update() (:463)
        A.check(L.srganfd_l1_loss(sr.data_ptr(), gt.data_ptr(), sr.numel(), self.pw, self.scalars.data_ptr() + 8, 0, dsr.data_ptr(), self.pw,
                                  self.scaler.seed_ptr, self.ws.data_ptr(), st), "l1_loss")
        adv_out = de.forward(sr, True)                      # updated D, SN state advances again (train_bsrgan.py:452)
        self._bce(adv_out, 1.0, self.aw, 3, None, dl)
        if self.train_generator:
            _, dsr_adv = de.backward(de._last, de.token, dl, False, True)
            A.check(L.srganfd_axpby(A.View(dsr_adv.data_ptr(), 1, 0), A.View(dsr.data_ptr(), 1, 0), A.F32, dsr.numel(), 1, 1.0, 1.0, st), "axpby")
            self.g_reducer.begin()
            gg, _ = ge.backward(g_sp, g_tok, dsr, False, on_ready=self.g_reducer.bucket)
            self.scaler.step(self.g_opt, gg, self.g_reducer.finish())        # scaler.step(g_optimizer); scaler.update(); EMA  (:466-470)
            ge.fp.touch()
        self.sr = sr
        return self.scalars
