"""Fused training iterations mirroring the reference's ``train()`` bodies, on the HIP engines.

  * ``GeneratorTrainer.step``  = ESRGAN/train_rrdbnet.py:244-272 / BSRGAN/train_bsrnet.py:244-272
    (zero_grad, G forward, L1 * weight, backward, Adam step, EMA update).
  * ``GanTrainer.step``        = BSRGAN/train_bsrgan.py:387-483 (see engine_d.py / gan.py).

They call the same engines as the ``nn.Module`` surface (``model.py``) but skip the autograd
bookkeeping: loss + dLoss/dSR come from one HIP kernel, the flat gradient goes straight to the fused
Adam + EMA kernel, and under data parallelism the flat gradient is all-reduced once (RCCL).
Mixed precision: the reference runs ``amp.autocast()`` + ``GradScaler`` (fp16 on CUDA, train_bsrgan.py:109,415-467).
``compute_dtype = torch.float16`` is that mode here -- f16 activations, packed weights and gradients, fp32 master weights and
accumulation -- with ``LossScaler`` below playing GradScaler's part (scaled loss seeds, unscale folded into the Adam kernel,
found-inf check on the flat gradient, skipped step, dynamic scale).  bf16 / f32 need no loss scaling: the scaler is then the
identity, as GradScaler is on the reference's CPU path.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from . import _abi as A
from .engine import generator_engine
from .parallel import BucketReducer, allreduce_sum_


class LossScaler:
    """``torch.cuda.amp.GradScaler()`` as the reference uses it (one instance for both networks, ``scale(loss).backward()``,
    ``step(optimizer)``, ``update()`` after EACH optimizer step: train_bsrgan.py:109,420,430,436-437,463,466-467), defaults of
    torch/amp/grad_scaler.py: init_scale 65536, growth x2 after 2000 consecutive finite steps, backoff x0.5 on a non-finite one.

    Like torch's, the scale lives in DEVICE memory (``state``: scale, 1 / scale, growth tracker, step and skip counts): the loss
    kernels multiply their gradient seeds by ``state[0]`` (``seed_ptr``), the Adam kernel unscales with ``state[1]`` and skips on the
    found_inf flag, and ``srganfd_loss_scale_update`` applies GradScaler.update()'s rule right behind the optimizer step, in stream
    order.  So the backward pass that follows an overflow already runs at the halved scale -- one skipped step per overflow, as in
    torch -- the host never waits for a flag, a captured graph never carries a stale scale, and under data parallelism every rank
    updates its copy from the same flag (found_inf is computed on the all-reduced gradient).  The loss is never materialised scaled.
    ``tests/test_gan_gpu.py::test_loss_scaler_follows_torch_gradscaler`` replays an overflow / growth sequence through both."""

    def __init__(self, device, enabled: bool = True, init_scale: float = 65536.0, growth_factor: float = 2.0, backoff_factor: float = 0.5,
                 growth_interval: int = 2000):
        self.enabled = enabled
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval
        if enabled:
            self.state = torch.zeros(8, dtype=torch.float32, device=device)
            self.state[0] = float(init_scale)
            self.state[1] = 1.0 / float(init_scale)
            self.flag = torch.zeros(1, dtype=torch.float32, device=device)

    # -- what the kernels read ------------------------------------------------------------------------------------------
    @property
    def seed_ptr(self):
        """device address of the scale: multiplies the gradient seed of a loss kernel (``grad_scale_dev``); None when disabled"""
        return self.state.data_ptr() if self.enabled else None

    @property
    def inv_ptr(self):
        return self.state.data_ptr() + 4 if self.enabled else None

    # -- host views (each one synchronises: reports, tests, checkpoints) ------------------------------------------------
    @property
    def scale(self) -> float:
        return float(self.state[0].item()) if self.enabled else 1.0

    @scale.setter
    def scale(self, value: float) -> None:
        if self.enabled:
            self.state[0] = float(value)
            self.state[1] = 1.0 / float(value)

    def current(self) -> float:
        """The scale the next backward pass will be seeded with (host copy: synchronises; the training loops do not call it)."""
        return self.scale

    def step(self, opt: "FlatAdamEMA", grad: Tensor, grad_scale: float, update_ema: bool = True) -> None:
        """GradScaler.step(optimizer) + update(): found_inf on the (all-reduced) flat gradient, unscale inside the (skipped) Adam
        step, then the scale update -- all on the device, in stream order."""
        if not self.enabled:
            opt.step(grad, grad_scale, update_ema)
            return
        L, st = A.lib(), A.stream_ptr()
        A.check(L.srganfd_nonfinite_flag(grad.data_ptr(), grad.numel(), self.flag.data_ptr(), 0, st), "nonfinite_flag")
        opt.step(grad, grad_scale, update_ema, skip_flag=self.flag, grad_scale_dev=self.inv_ptr)
        A.check(L.srganfd_loss_scale_update(self.state.data_ptr(), self.flag.data_ptr(), self.growth_factor, self.backoff_factor,
                                            int(self.growth_interval), st), "loss_scale_update")

    def report(self) -> dict:
        if not self.enabled:
            return {"enabled": False, "scale": 1.0, "optimizer_steps": 0, "skipped": 0}
        st = self.state.cpu()
        cnt = st.view(torch.int32)                 # words 2..4 are int32 counters (a float would stop counting at 2^24 steps)
        return {"enabled": True, "scale": float(st[0]), "optimizer_steps": int(cnt[3]), "skipped": int(cnt[4])}

    @property
    def n_steps(self) -> int:
        return self.report()["optimizer_steps"]

    @property
    def n_skipped(self) -> int:
        return self.report()["skipped"]

    def state_dict(self) -> dict:
        """torch GradScaler.state_dict() keys (the reference does not checkpoint its scaler; kept for symmetry)."""
        st = self.state.cpu() if self.enabled else None
        return {"scale": float(st[0]) if st is not None else 1.0, "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval, "_growth_tracker": int(st.view(torch.int32)[2]) if st is not None else 0}

    def load_state_dict(self, sd: dict) -> None:
        """torch GradScaler.load_state_dict: scale, tracker and the three hyper-parameters state_dict() saves"""
        self.growth_factor = float(sd.get("growth_factor", self.growth_factor))
        self.backoff_factor = float(sd.get("backoff_factor", self.backoff_factor))
        self.growth_interval = int(sd.get("growth_interval", self.growth_interval))
        if self.enabled:
            sc = float(sd["scale"])
            self.state[0] = sc
            self.state[1] = 1.0 / sc
            self.state.view(torch.int32)[2] = int(sd.get("_growth_tracker", 0))


class MultiStepLR:
    """``torch.optim.lr_scheduler.MultiStepLR`` for the fused optimizers (torch's class insists on a ``torch.optim.Optimizer``).
    The reference builds one per network from ``lr_scheduler_milestones`` / ``lr_scheduler_gamma`` (bsrgan_config.py:153-155,
    train_bsrgan.py:314-323) and steps it once per epoch (:193-195).  ``state_dict()`` carries torch's keys, so a checkpoint written
    by the reference's script resumes here and the other way round (utils.load_state_dict passes ``ckpt["scheduler"]`` through)."""

    def __init__(self, optimizer: "FlatAdamEMA", milestones, gamma: float = 0.1, last_epoch: int = -1):
        from collections import Counter
        self.optimizer = optimizer
        self.milestones = Counter(int(m) for m in milestones)
        self.gamma = gamma
        self.base_lrs = [optimizer.lr]
        self.last_epoch = last_epoch
        self._step_count = 0
        self._last_lr = [optimizer.lr]
        self.step()                                   # torch's constructor performs the initial step (last_epoch -> 0)

    def get_last_lr(self):
        return list(self._last_lr)

    def step(self) -> None:
        self._step_count += 1
        self.last_epoch += 1
        if self.last_epoch in self.milestones:        # torch's chained form: multiply the CURRENT rate (a rate set by hand is kept)
            self.optimizer.lr = self.optimizer.lr * self.gamma ** self.milestones[self.last_epoch]
        self._last_lr = [self.optimizer.lr]

    def state_dict(self) -> dict:
        return {"milestones": self.milestones, "gamma": self.gamma, "base_lrs": list(self.base_lrs), "last_epoch": self.last_epoch,
                "verbose": False, "_step_count": self._step_count, "_get_lr_called_within_step": False, "_last_lr": list(self._last_lr)}

    def load_state_dict(self, sd: dict) -> None:
        from collections import Counter
        self.milestones = Counter({int(k): int(v) for k, v in dict(sd["milestones"]).items()})
        self.gamma, self.base_lrs = sd["gamma"], list(sd["base_lrs"])
        self.last_epoch, self._step_count = int(sd["last_epoch"]), int(sd["_step_count"])
        self._last_lr = list(sd["_last_lr"])
        self.optimizer.lr = self._last_lr[0]


def pin_training_dtype(*modules) -> torch.dtype:
    """The fused trainers ARE the reference's training loops, and those run every forward under ``amp.autocast()`` (train_bsrgan.py:415-427,
    450-457; train_bsrnet.py:252-254): float16 + GradScaler.  A module whose ``compute_dtype`` is still None (the default: follow autocast)
    is therefore pinned here, when a trainer takes it over, to the dtype of the autocast region the trainer is built in or -- outside any --
    to the reference loops' float16.  Without this a trainer built as INTEGRATION.md shows (``GanTrainer(g, d, cl)``, outside autocast) would
    train in exact fp32: the parity kernels, twice the activation memory, no thin-side kernels, no loss scaling -- nothing the reference's
    loop does.  An explicit ``compute_dtype`` (float32 for the parity tests and configs[0], bfloat16) is kept."""
    dt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else torch.float16
    for m in modules:
        if m is not None and getattr(m, "compute_dtype", None) is None:
            m.compute_dtype = dt
    return dt


def needs_loss_scaling(*modules) -> bool:
    """float16 is the one mode that needs GradScaler's part played: an explicit ``compute_dtype`` or -- the default -- the dtype of
    the autocast region the trainer is built and stepped in (engine.resolve_compute_dtype)."""
    from .engine import resolve_compute_dtype
    return any(resolve_compute_dtype(m) == torch.float16 for m in modules if m is not None)


def check_loss_scaling(scaler: "LossScaler", *modules) -> None:
    """``compute_dtype`` is a plain attribute of the modules (and autocast a thread-local state): switched to float16 AFTER the trainer
    was built, training would run f16 with the scaler disabled and the gradients would silently underflow.  Checked at every step
    (an attribute read per module)."""
    if not scaler.enabled and needs_loss_scaling(*modules):
        raise A.SrganfdError("compute_dtype was set to torch.float16 (or a float16 autocast region entered) after the trainer was built: "
                             "its loss scaler is disabled (gradients would underflow) -- set the dtype first, then build the trainer")


class FlatAdamEMA:
    """torch.optim.Adam maths (amsgrad=False) + AveragedModel(avg_fn=(1-d)*ema + d*p) over one flat buffer."""

    def __init__(self, flat: Tensor, lr: float, betas: Tuple[float, float], eps: float, weight_decay: float = 0.0,
                 ema_decay: Optional[float] = None, layout=None):
        """``layout``: the engine's FlatParams (names / offsets / shapes of the tensors inside ``flat``); needed only by
        state_dict() / load_state_dict(), which speak the reference checkpoint's per-parameter format."""
        self.layout = layout
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

    # -- checkpoints in the reference's format (train_bsrnet.py:124-130: {"optimizer": Adam.state_dict(), "ema_state_dict": ...}) --
    def _views(self, buf: Tensor):
        lay = self.layout
        if lay is None:
            raise A.SrganfdError("FlatAdamEMA: state_dict needs the parameter layout (pass layout=engine.fp)")
        return [buf[o:o + n].view(s) for o, n, s in zip(lay.offsets, lay.numels, lay.shapes)]

    def state_dict(self) -> dict:
        """What ``torch.optim.Adam(module.parameters(), ...).state_dict()`` holds after the same steps: per parameter (in
        named_parameters() order) ``step`` / ``exp_avg`` / ``exp_avg_sq``, and one param group."""
        m, v = self._views(self.m), self._views(self.v)
        state = {}
        t = int(self.step_dev.item()) if self.step_dev is not None else self.t      # skipped (non-finite) steps count on the device only
        if t > 0:
            state = {i: {"step": torch.tensor(float(t)), "exp_avg": m[i].detach().clone(), "exp_avg_sq": v[i].detach().clone()}
                     for i in range(len(m))}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.wd, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None, "params": list(range(len(m)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd: dict) -> None:
        """Accepts a checkpoint's ``optimizer`` entry written by torch.optim.Adam over the same module (or by state_dict() above).
        Restores moments, step count (host and, if in use, device copy) and the hyper-parameters of the param group."""
        m, v = self._views(self.m), self._views(self.v)
        st = sd.get("state", {})
        if len(st) not in (0, len(m)):
            raise A.SrganfdError(f"optimizer state holds {len(st)} parameters, the module has {len(m)}")
        steps = set()
        with torch.no_grad():
            self.m.zero_()
            self.v.zero_()
            for i in range(len(m)):
                e = st.get(i, st.get(str(i)))
                if e is None:
                    continue
                m[i].copy_(e["exp_avg"].reshape(m[i].shape))
                v[i].copy_(e["exp_avg_sq"].reshape(v[i].shape))
                steps.add(int(float(e["step"])))
        if len(steps) > 1:
            raise A.SrganfdError(f"parameters carry different step counts {sorted(steps)}: not a state the fused optimizer can resume")
        self.t = steps.pop() if steps else 0
        if self.step_dev is not None:
            self.step_dev.fill_(self.t)
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps, self.wd = float(g["lr"]), tuple(g["betas"]), float(g["eps"]), float(g.get("weight_decay", 0.0))

    def ema_state_dict(self) -> dict:
        """``AveragedModel(module).state_dict()`` layout (torch/optim/swa_utils.py): ``module.<name>`` tensors + ``n_averaged``."""
        if self.ema is None:
            raise A.SrganfdError("this optimizer keeps no EMA copy")
        out = {"n_averaged": torch.tensor(self.n_averaged, dtype=torch.long)}
        src = self._views(self.ema if self.n_averaged > 0 else self.flat)      # AveragedModel starts as a deep copy of the module
        for name, t in zip(self.layout.names, src):
            out["module." + name] = t.detach().clone()
        return out

    def load_ema_state_dict(self, sd: dict) -> None:
        if self.ema is None:
            raise A.SrganfdError("this optimizer keeps no EMA copy")
        dst = self._views(self.ema)
        with torch.no_grad():
            for name, t in zip(self.layout.names, dst):
                src = sd.get("module." + name)
                if src is not None and tuple(src.shape) == tuple(t.shape):
                    t.copy_(src)
        self.n_averaged = int(sd.get("n_averaged", 0))

    def step(self, grad: Tensor, grad_scale: float = 1.0, update_ema: bool = True, skip_flag: Optional[Tensor] = None,
             grad_scale_dev: Optional[int] = None) -> None:
        """``skip_flag`` (device float, LossScaler): non-zero skips the update on the device; the step count then has to live on
        the device too (a skipped step does not advance torch's ``state["step"]``), so the device-step kernel is used.
        ``grad_scale_dev``: device address of a float that multiplies ``grad_scale`` (1 / loss scale)."""
        if skip_flag is not None:
            self.use_device_step()
        if self.layout is not None and self.layout.flat is not self.flat:
            raise A.SrganfdError("the module's flat parameter buffer was rebuilt (.to() / deepcopy) after this optimizer captured it: "
                                 "build the trainer after moving the module")
        self.t += 1
        mode = 0
        if self.ema is not None and update_ema:
            mode = 1 if self.n_averaged == 0 else 2
            self.n_averaged += 1
        if self.step_dev is not None:
            A.check(A.lib().srganfd_adam_ema_dev(self.flat.data_ptr(), grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                                 self.ema.data_ptr() if self.ema is not None else None, self.flat.numel(), self.lr,
                                                 self.betas[0], self.betas[1], self.eps, self.wd, self.step_dev.data_ptr(),
                                                 self.bc_dev.data_ptr(), grad_scale, self.ema_decay or 0.0, mode,
                                                 skip_flag.data_ptr() if skip_flag is not None else None, grad_scale_dev, A.stream_ptr()), "adam_ema_dev")
            return
        A.check(A.lib().srganfd_adam_ema(self.flat.data_ptr(), grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                         self.ema.data_ptr() if self.ema is not None else None, self.flat.numel(), self.lr,
                                         self.betas[0], self.betas[1], self.eps, self.wd, self.t, grad_scale,
                                         self.ema_decay or 0.0, mode, None, grad_scale_dev, A.stream_ptr()), "adam_ema")


class GanCheckpointMixin:
    """state_dict() / load_state_dict() of the two-network trainers (gan.GanTrainer, gan_esrgan.EsrganGanTrainer): the trainer-side
    entries of the reference's two checkpoint files (train_bsrgan.py:203-260, ESRGAN/train_esrgan.py:216-262: d_*.pth.tar holds the
    discriminator + its optimizer, g_*.pth.tar the generator + optimizer + EMA), keyed "g" / "d"; "scaler" is an addition (the
    reference does not checkpoint its GradScaler and restarts it at 65536).  Needs self.g / .d / .ge / .de / .g_opt / .d_opt / .scaler."""

    def state_dict(self) -> dict:
        g = {"state_dict": self.g.state_dict(), "optimizer": self.g_opt.state_dict()}
        if self.g_opt.ema is not None:
            g["ema_state_dict"] = self.g_opt.ema_state_dict()
        return {"g": g, "d": {"state_dict": self.d.state_dict(), "optimizer": self.d_opt.state_dict()}, "scaler": self.scaler.state_dict()}

    def load_state_dict(self, ckpt: dict) -> None:
        with torch.no_grad():
            for net, eng, opt, c in ((self.g, self.ge, self.g_opt, ckpt.get("g")), (self.d, self.de, self.d_opt, ckpt.get("d"))):
                if c is None:
                    continue
                own = net.state_dict()
                for k, v in c["state_dict"].items():        # spectral-norm u / v and BatchNorm running statistics included
                    if k in own and tuple(own[k].shape) == tuple(v.shape):
                        own[k].copy_(v)
                eng.fp.touch()
                if "optimizer" in c:
                    opt.load_state_dict(c["optimizer"])
                if "ema_state_dict" in c and opt.ema is not None:
                    opt.load_ema_state_dict(c["ema_state_dict"])
        if "scaler" in ckpt:
            self.scaler.load_state_dict(ckpt["scaler"])


class GeneratorTrainer:
    """Generator-only iteration (BASELINE.json configs[0] and [1])."""

    def __init__(self, g_model, lr: float, betas=(0.9, 0.99), eps: float = 1e-8, weight_decay: float = 0.0,
                 ema_decay: Optional[float] = 0.999, loss_weight: float = 1.0, process_group=None):
        self.g = g_model
        pin_training_dtype(g_model)
        self.eng = generator_engine(g_model)
        dev = next(g_model.parameters()).device
        self.flat = self.eng.fp.sync(dev)
        self.opt = FlatAdamEMA(self.flat, lr, betas, eps, weight_decay, ema_decay, layout=self.eng.fp)
        self.loss_weight = loss_weight
        self.scaler = LossScaler(dev, enabled=needs_loss_scaling(g_model))      # train_rrdbnet.py:94 / train_bsrnet.py:94
        self.pg = process_group
        self.g_reducer = BucketReducer(dev, process_group)
        self.loss_buf = torch.zeros(1, dtype=torch.float32, device=dev)
        self.ws = torch.empty(A.LOSS_WS_FLOATS, dtype=torch.float32, device=dev)
        self.dsr: Optional[Tensor] = None

    def state_dict(self) -> dict:
        """The entries of the reference's checkpoint that belong to the trainer (train_bsrnet.py:124-130)."""
        out = {"state_dict": self.g.state_dict(), "optimizer": self.opt.state_dict()}
        if self.opt.ema is not None:
            out["ema_state_dict"] = self.opt.ema_state_dict()
        return out

    def load_state_dict(self, ckpt: dict) -> None:
        """Resume from a checkpoint written by the reference's train script or by state_dict(): weights (in place, into the
        flat buffer), Adam moments + step, EMA copy + n_averaged."""
        with torch.no_grad():
            own = self.g.state_dict()
            for k, v in ckpt["state_dict"].items():
                if k in own and tuple(own[k].shape) == tuple(v.shape):
                    own[k].copy_(v)
        self.eng.fp.touch()
        if "optimizer" in ckpt:
            self.opt.load_state_dict(ckpt["optimizer"])
        if "ema_state_dict" in ckpt and self.opt.ema is not None:
            self.opt.load_ema_state_dict(ckpt["ema_state_dict"])

    def step(self, lr_img: Tensor, gt: Tensor) -> Tensor:
        """Returns the (device, 1-element) loss tensor; no host sync inside."""
        eng = self.eng
        sr = eng.forward(lr_img, True)
        sp, token = eng._last, eng.token
        if self.dsr is None or self.dsr.shape != sr.shape:
            self.dsr = torch.empty_like(sr)
        gt = gt.contiguous().float()
        check_loss_scaling(self.scaler, self.g)
        # scaler.scale(loss): the (device-resident) factor rides on the gradient seed
        A.check(A.lib().srganfd_l1_loss(sr.data_ptr(), gt.data_ptr(), sr.numel(), self.loss_weight, self.loss_buf.data_ptr(), 0,
                                        self.dsr.data_ptr(), self.loss_weight, self.scaler.seed_ptr, self.ws.data_ptr(), A.stream_ptr()), "l1_loss")
        # RCCL over xGMI: the flat gradient goes out in three buckets as the backward pass finishes them (parallel.BucketReducer)
        self.g_reducer.begin()
        grad, _ = eng.backward(sp, token, self.dsr, False, on_ready=self.g_reducer.bucket)
        scale = self.g_reducer.finish()
        self.scaler.step(self.opt, grad, scale)        # scaler.step(optimizer); scaler.update(); ema update
        eng.fp.touch()                             # parameters changed behind autograd's back -> re-pack
        self.sr = sr
        return self.loss_buf
