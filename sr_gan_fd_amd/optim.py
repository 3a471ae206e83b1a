"""Drop-in ``Adam`` for the reference's train scripts (``optim.Adam(model.parameters(), lr, betas, eps, weight_decay)``,
train_bsrgan.py:311-323; train_bsrnet.py:184-189): torch.optim.Adam's interface, state and state_dict, with the update of a whole
network as ONE HIP kernel (srganfd_adam_ema) whenever the parameters and their gradients are views of one flat buffer each -- which
is how this package's engines lay them out.  Anything else (other modules, amsgrad / maximize, a parameter without gradient, gradients
autograd had to copy) runs torch's own step on the same state tensors.

    - from torch import optim                      # the scripts' import
    + from sr_gan_fd_amd import optim              # optim.Adam; lr_scheduler is torch's (re-exported)

Measured on the module-level BSRGAN loop (batch 32, 128 -> 512, bench.py --module-loop): GradScaler.step + torch's foreach Adam over
702 tensors 4.3 ms per optimizer step; this class: one 0.1 ms launch behind the scaler's unscale.
"""
from __future__ import annotations

import torch
from torch.optim import lr_scheduler  # noqa: F401  (the scripts use optim.lr_scheduler.MultiStepLR)

from . import _abi as A
from .flat import flat_span


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **kw):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, **kw)
        self.flat_steps = 0                      # steps taken on the fused path (tests / reports)
        self._flat = {}                          # group index -> dict(t=step count, m=flat exp_avg, v=flat exp_avg_sq, sig=layout signature)

    # -- fused path ----------------------------------------------------------------------------------------------------------
    def _group_flat(self, gi: int, group) -> bool:
        if group.get("amsgrad") or group.get("maximize") or group.get("capturable") or group.get("differentiable") or isinstance(group["lr"], torch.Tensor):
            return False
        # fused=True makes GradScaler.step skip unscale_ and its inf check and hand grad_scale / found_inf to the optimizer as
        # attributes (torch/amp/grad_scaler.py: _step_supports_amp_scaling): the flat kernel below reads neither, so it would apply
        # loss-scaled gradients and never skip a non-finite step.  That contract is torch's fused implementation's: let it run.
        if group.get("fused") or getattr(self, "grad_scale", None) is not None or getattr(self, "found_inf", None) is not None:
            return False
        ps = group["params"]
        if not ps or not ps[0].is_cuda:
            return False
        pf = flat_span([p.data for p in ps])
        gf = flat_span([p.grad for p in ps])
        if pf is None or gf is None or pf[1] != gf[1] or pf[0].numel() != gf[0].numel():
            return False
        flat, offs = pf
        sig = (flat.data_ptr(), flat.numel(), tuple(offs))
        st = self._flat.get(gi)
        if st is None or st["sig"] != sig:
            # (re)build the flat moments from whatever per-parameter state exists, and make the per-parameter state views of them
            m, v = torch.zeros_like(flat), torch.zeros_like(flat)
            steps = set()
            for p, o in zip(ps, offs):
                s = self.state.get(p)
                if s and "exp_avg" in s:
                    m[o:o + p.numel()].copy_(s["exp_avg"].reshape(-1))
                    v[o:o + p.numel()].copy_(s["exp_avg_sq"].reshape(-1))
                    steps.add(int(float(s["step"])))
                else:
                    steps.add(0)
            if len(steps) > 1:
                return False                     # parameters at different step counts (partial state): one bias correction cannot serve them
            t = steps.pop() if steps else 0
            for p, o in zip(ps, offs):
                s = self.state[p]
                s["exp_avg"], s["exp_avg_sq"] = m[o:o + p.numel()].view_as(p), v[o:o + p.numel()].view_as(p)
                s["step"] = torch.tensor(float(t))
            st = self._flat[gi] = dict(t=t, m=m, v=v, sig=sig)
        st["flat"], st["grad"] = flat, gf[0]
        return True

    def _sync_steps(self) -> None:
        """write the fused path's step counts into the per-parameter state (what torch's code and state_dict() read)"""
        for gi, group in enumerate(self.param_groups):
            st = self._flat.get(gi)
            if st is not None:
                for p in group["params"]:
                    if p in self.state:
                        self.state[p]["step"] = torch.tensor(float(st["t"]))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if not all(self._group_flat(gi, g) for gi, g in enumerate(self.param_groups)):
            self._sync_steps()
            self._flat.clear()                   # torch's step advances the per-parameter state: rebuild from it next time
            super().step()
            return loss
        L = A.lib()
        for gi, group in enumerate(self.param_groups):
            st = self._flat[gi]
            st["t"] += 1
            b1, b2 = group["betas"]
            A.check(L.srganfd_adam_ema(st["flat"].data_ptr(), st["grad"].data_ptr(), st["m"].data_ptr(), st["v"].data_ptr(), None, st["flat"].numel(),
                                       float(group["lr"]), b1, b2, group["eps"], group["weight_decay"], st["t"], 1.0, 0.0, 0, None, None, A.stream_ptr()), "adam")
        self._bump_versions()
        self.flat_steps += 1
        return loss

    def _bump_versions(self) -> None:
        """The kernel wrote the parameters behind autograd's back: the engines compare tensor version counters to know when to re-pack
        their weights (FlatParams.signature), so bump them the way an in-place torch op would."""
        for group in self.param_groups:
            torch.autograd.graph.increment_version(group["params"])

    def state_dict(self):
        self._sync_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        self._flat.clear()
        return super().load_state_dict(state_dict)
