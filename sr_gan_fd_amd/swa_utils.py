"""Drop-in ``AveragedModel`` for the reference's EMA (``AveragedModel(g_model, avg_fn=ema_avg)``, train_bsrgan.py:290-291,470; the same in
train_bsrnet.py / the ESRGAN scripts): torch.optim.swa_utils.AveragedModel with ``update_parameters`` applied to the whole network at once.

torch's implementation calls ``avg_fn`` once per parameter -- 702 calls for the 23-block RRDBNet, each a handful of tiny kernels plus a
device copy of ``n_averaged`` -- 26.6 ms of the module-level BSRGAN step (bench.py --module-loop: 92 ms against the fused trainer's 59).
The scripts' ``avg_fn`` is elementwise tensor arithmetic, so applying it ONCE to the flat buffers the engines keep the parameters in gives
every element the same value: the same three kernels over 67 MB instead of over 702 tensors.  **Requirement: ``avg_fn`` / ``multi_avg_fn``
must be elementwise** (the reference's ``(1 - decay) * averaged + decay * current``, torch's ``get_ema_avg_fn`` / ``get_swa_avg_fn``); a
function that looks at a tensor as a whole (per-tensor norms, shape-dependent logic) would see the network as one 1-D tensor -- pass
``flat=False`` for such a function and torch's per-parameter loop runs instead.  Anything that is not laid out flat (other modules, CPU,
use_buffers=True) goes through torch's own loop as well.

    - from torch.optim.swa_utils import AveragedModel
    + from sr_gan_fd_amd.swa_utils import AveragedModel
"""
from __future__ import annotations

import torch
from torch.optim import swa_utils as _swa

from .flat import engine_flatten, flat_span


class AveragedModel(_swa.AveragedModel):
    def __init__(self, model, device=None, avg_fn=None, multi_avg_fn=None, use_buffers=False, flat=True):
        super().__init__(model, device=device, avg_fn=avg_fn, multi_avg_fn=multi_avg_fn, use_buffers=use_buffers)
        self.flat = flat
        self.flat_updates = 0                    # updates taken on the whole-network path (tests / reports)
        if flat and not use_buffers:
            # Parameter.__deepcopy__ clones every tensor, so the copy torch just made is 702 separate allocations: lay both networks
            # out flat once, here (the engines would do it at their first forward; the EMA copy may never run one before its first update)
            engine_flatten(model)
            engine_flatten(self.module)

    @torch.no_grad()
    def update_parameters(self, model) -> None:
        fa = fm = None
        if self.flat and not self.use_buffers:
            engine_flatten(model)                # no-ops (a pointer walk) unless a module was moved or rebuilt since __init__
            engine_flatten(self.module)
            pa, pm = list(self.module.parameters()), list(model.parameters())
            if len(pa) == len(pm) and all(a.shape == b.shape for a, b in zip(pa, pm)):
                fa, fm = flat_span([p.data for p in pa]), flat_span([p.data for p in pm])
        if fa is None or fm is None or fa[1] != fm[1] or fa[0].numel() != fm[0].numel() or fa[0].device != fm[0].device:
            return super().update_parameters(model)
        avg, src = fa[0], fm[0]
        if self.n_averaged == 0:
            avg.copy_(src)
        else:
            n = self.n_averaged.to(avg.device)
            if self.multi_avg_fn is not None:
                self.multi_avg_fn([avg], [src], n)
            elif self.avg_fn is not None:
                avg.copy_(self.avg_fn(avg, src, n))
            else:
                avg.copy_(_swa.get_swa_avg_fn()(avg, src, n))
        torch.autograd.graph.increment_version(pa)         # the copy's engine re-packs its weights when these change
        for b_swa, b_model in zip(self.module.buffers(), model.buffers()):      # buffers follow the source model (torch does the same)
            b_swa.detach().copy_(b_model.detach().to(b_swa.device))
        self.n_averaged += 1
        self.flat_updates += 1
