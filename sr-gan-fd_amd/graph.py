"""hipGraph replay of a fused training iteration.

A 23-block generator step is ~1400 kernel launches; at the benchmark batch (32 x 128x128) the GPU work hides the launch
cost, at small batches (BASELINE configs[0]: 4 x 32x32) the step is launch-bound.  ``GraphedStep`` captures one call of
``trainer.step`` (GeneratorTrainer or GanTrainer) into a hipGraph through ``torch.cuda.CUDAGraph`` and replays it:
the engines only launch kernels on torch's current stream (no host synchronisation, no host-side scalars that change
between iterations once the Adam step count lives in device memory), so the capture is exact.
Single-process only: the RCCL all-reduce of the data-parallel path is not captured here.
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
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = trainer.step(self.lr, self.gt)
        # the host-side counters advanced once during capture although nothing ran: undo
        for opt in (getattr(trainer, "opt", None), getattr(trainer, "g_opt", None), getattr(trainer, "d_opt", None)):
            if opt is not None:
                opt.t -= 1
                if opt.ema is not None:
                    opt.n_averaged -= 1

    def __call__(self, lr_img: Tensor, gt: Tensor) -> Tensor:
        self.lr.copy_(lr_img)
        self.gt.copy_(gt)
        self.graph.replay()
        for opt in (getattr(self.trainer, "opt", None), getattr(self.trainer, "g_opt", None), getattr(self.trainer, "d_opt", None)):
            if opt is not None:
                opt.t += 1
                if opt.ema is not None:
                    opt.n_averaged += 1
        return self.out
