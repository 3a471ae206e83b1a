"""hipGraph replay of a fused training iteration.

A 23-block generator step is ~1400 kernel launches; at the benchmark batch (32 x 128x128) the GPU work hides the launch
cost, at small batches (BASELINE configs[0]: 4 x 32x32) the step is launch-bound.  ``GraphedStep`` captures one call of
``trainer.step`` (GeneratorTrainer or GanTrainer) into a hipGraph through ``torch.cuda.CUDAGraph`` and replays it:
the engines only launch kernels on torch's current stream (no host synchronisation, no host-side scalars that change
between iterations once the Adam step count lives in device memory), so the capture is exact.
Single-process only: the RCCL all-reduce of the data-parallel path is not captured here.  f16 trainers (dynamic loss scaling,
trainer.LossScaler) need nothing special: the scale, the found_inf flag and GradScaler.update() all live on the device, so the
captured kernels read and advance them at every replay exactly as the eager step does.
"""
from __future__ import annotations

import torch
from torch import Tensor


class GraphedStep:
    def __init__(self, trainer, lr_example: Tensor, gt_example: Tensor, warmup: int = 2):
        """Runs ``warmup`` REAL iterations on the example batch (they build the launch plans and move the EMA past its
        first-call copy), then captures one more call without executing it."""
        if getattr(trainer, "pg", None) is not None:
            raise ValueError("GraphedStep: data-parallel trainers are not captured (the all-reduce stays eager)")
        self.trainer = trainer
        self.lr, self.gt = lr_example.clone(), gt_example.clone()      # static input buffers
        for opt in (getattr(trainer, "opt", None), getattr(trainer, "g_opt", None), getattr(trainer, "d_opt", None)):
            if opt is not None:
                opt.use_device_step()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                trainer.step(self.lr, self.gt)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._capture()

    def _opts(self):
        return [o for o in (getattr(self.trainer, "opt", None), getattr(self.trainer, "g_opt", None), getattr(self.trainer, "d_opt", None)) if o is not None]

    def _capture(self) -> None:
        """Kernel arguments are frozen at capture: learning rate, betas, eps and the loss weights are by-value arguments of the
        Adam / loss kernels (only the step count lives in device memory).  They are recorded here and checked at every replay;
        a change (the reference steps MultiStepLR every epoch, train_bsrgan.py:193-195) re-captures the graph."""
        opts = self._opts()
        before = [(o.t, o.n_averaged) for o in opts]
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = self.trainer.step(self.lr, self.gt)
        # the host-side counters advanced during capture although nothing ran: undo exactly what moved (an optimizer that
        # did not step -- train_generator=False -- keeps its counters)
        self._delta = []
        for o, (t0, n0) in zip(opts, before):
            self._delta.append((o.t - t0, o.n_averaged - n0))
            o.t, o.n_averaged = t0, n0
        self._frozen = self._hyper()

    def _hyper(self):
        tr = self.trainer
        return ([(o.lr, tuple(o.betas), o.eps, o.wd, o.ema_decay) for o in self._opts()],
                tuple(repr(getattr(tr, k, None)) for k in ("pw", "cw", "aw", "loss_weight", "train_generator")))

    def __call__(self, lr_img: Tensor, gt: Tensor) -> Tensor:
        if self._hyper() != self._frozen:
            self._capture()
        self.lr.copy_(lr_img)
        self.gt.copy_(gt)
        self.graph.replay()
        for o, (dt, dn) in zip(self._opts(), self._delta):
            o.t += dt
            o.n_averaged += dn
        return self.out
