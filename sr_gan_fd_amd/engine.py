"""Execution engines: sequence the HIP kernels (C ABI) for the reference's modules.

Host code here is plumbing only: it owns device buffers (torch tensors), builds the launch argument
structs once per input shape, and replays them on torch's current stream.  All arithmetic is in
libsrganfd_hip.so; nothing in this file falls back to PyTorch ops for compute.

Data layout in HBM (per engine, per input shape):
  * parameters: ONE flat fp32 buffer in ``named_parameters()`` order; every ``nn.Parameter`` of the
    module is re-pointed to a view of it (so optimizers / state_dict see the reference's tensors,
    while Adam, EMA and the gradient all-reduce work on one contiguous range).
  * packed weights: bf16 (or f32) MFMA B-fragment order, forward and data-gradient orientation,
    re-packed by one kernel launch whenever the flat parameter buffer changed.
  * activations: NHWC.  Each dense block owns ONE (N,H,W,C+4G) buffer: channels [0,C) are the
    block input, [C+kG, C+(k+1)G) the k-th growth conv's output -- torch.cat never runs, and the
    block's output is written by conv5's epilogue straight into the next block's channels [0,C).
  * backward: per dense block ONE stacked-gradient buffer [dOut(C) | dY4 | dY3 | dY2 | dY1]; the
    data-gradient of the block is again a dense block over that buffer (5 launches, no
    read-modify-write), and all five weight gradients come from one launch.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch import Tensor, nn

from . import _abi as A
from . import ops
from . import profiling

_ENGINES: "weakref.WeakKeyDictionary[nn.Module, object]" = weakref.WeakKeyDictionary()


def _require_gpu(x: Tensor) -> None:
    if not x.is_cuda and not A.DRY_RUN:
        raise A.SrganfdError("sr_gan_fd_amd runs on MI355X only: input tensor is not on a GPU (no CPU fallback)")
    A.lib()


def resolve_compute_dtype(module) -> torch.dtype:
    """The reference's precision contract for its modules: a forward issued inside ``torch.autocast("cuda")`` (the training loops'
    ``amp.autocast()``, train_bsrgan.py:415-427,450-457; ESRGAN's validation too, train_rrdbnet.py:324-325) computes in the autocast
    dtype -- float16 by default, what the reference's convs run in on a GPU -- and one issued outside it (``validate()``,
    train_bsrgan.py:563; inference.py:60-70) in float32.  ``module.compute_dtype`` (None by default) overrides both.  The backward
    pass always runs in the dtype of the forward whose activations it consumes (the plan's), as autograd does under autocast."""
    dt = getattr(module, "compute_dtype", None)
    if dt is not None:
        return dt
    if torch.is_autocast_enabled("cuda"):
        return torch.get_autocast_dtype("cuda")
    return torch.float32


def _dt(module) -> Tuple[torch.dtype, int]:
    dt = resolve_compute_dtype(module)
    if dt not in ops.DT:
        raise A.SrganfdError(f"compute dtype {dt} is not supported (float32, float16, bfloat16)")
    return dt, ops.DT[dt]


# ------------------------------------------------------------------------------------------------
# flat parameter storage
# ------------------------------------------------------------------------------------------------
class FlatParams:
    """Keeps a module's parameters as views of one contiguous fp32 buffer."""

    def __init__(self, named: Sequence[Tuple[str, nn.Parameter]]):
        self.names = [n for n, _ in named]
        self.params = [p for _, p in named]
        self.shapes = [tuple(p.shape) for p in self.params]
        self.numels = [p.numel() for p in self.params]
        self.offsets, off = [], 0
        for n in self.numels:
            self.offsets.append(off)
            off += (n + 3) // 4 * 4          # keep every tensor 16-byte aligned
        self.total = off
        self.index = {n: i for i, n in enumerate(self.names)}
        self.flat: Optional[Tensor] = None
        self.version = 0                       # bumped by whoever writes the flat buffer behind autograd's back (touch())

    def off(self, name: str) -> int:
        return self.offsets[self.index[name]]

    def sync(self, device) -> Tensor:
        """(Re)build the flat buffer if any parameter moved (``.to()``, deepcopy, fresh module)."""
        f = self.flat
        ok = f is not None and f.device == device
        if ok:
            base = f.data_ptr()
            for p, o in zip(self.params, self.offsets):
                if p.data_ptr() != base + 4 * o:
                    ok = False
                    break
        if not ok:
            f = torch.zeros(self.total, dtype=torch.float32, device=device)
            with torch.no_grad():
                for p, o, n, s in zip(self.params, self.offsets, self.numels, self.shapes):
                    f[o:o + n].copy_(p.detach().reshape(-1).to(device=device, dtype=torch.float32))
                    p.data = f[o:o + n].view(s)
            self.flat = f
            self.version += 1
        return f

    def touch(self) -> None:
        """The flat buffer was written by a kernel (fused Adam): every packed copy, of every compute dtype, is stale."""
        self.version += 1

    def signature(self) -> tuple:
        """Identity of the current parameter VALUES (tensor version counters + touch() count)."""
        return (self.flat.data_ptr(), self.flat._version, self.version, tuple(p._version for p in self.params))

    def stale(self, pk: dict) -> bool:
        """True if the packed-weight record `pk` (one per compute dtype) was packed from other values; marks it fresh.
        The signature is kept PER PACK: with one shared marker, the first forward in dtype A after an optimizer step consumed the
        change and a following forward in dtype B ran on weights from before the step."""
        sig = self.signature()
        if pk.get("seen") != sig:
            pk["seen"] = sig
            return True
        return False

    def new_grad(self, device) -> Tensor:
        """Flat gradient buffer.  Tensors whose size is not a multiple of 4 leave alignment gaps that no kernel writes: zero
        them (the optimizer walks the whole buffer), otherwise the gaps of the parameter buffer pick up allocator garbage and
        two identical runs stop being bitwise equal (found by tests/test_fullsize_gpu.py)."""
        if self.total == sum(self.numels):
            return torch.empty(self.total, dtype=torch.float32, device=device)
        return torch.zeros(self.total, dtype=torch.float32, device=device)

    def grad_views(self, flat_grad: Tensor) -> List[Tensor]:
        return [flat_grad[o:o + n].view(s) for o, n, s in zip(self.offsets, self.numels, self.shapes)]


# ------------------------------------------------------------------------------------------------
# generator (RRDBNet / BSRGAN) engine
# ------------------------------------------------------------------------------------------------
class _Shape:
    """Per-(N,H,W,training) buffers and pre-built launch lists."""
    pass


class PlanCache:
    """Per-shape plans, least-recently-used eviction.  Training plans are pinned (at most two are kept): a validation sweep over
    many image sizes (inference.py / validate()) evicts only other inference plans, never the multi-GB training
    plan of the step it runs between -- the earlier `clear()` on overflow dropped everything."""

    def __init__(self, cap: int = 8, cap_pinned: int = 2):
        from collections import OrderedDict
        self.d: "OrderedDict[tuple, tuple]" = OrderedDict()
        self.cap, self.cap_pinned = cap, cap_pinned

    def get(self, key):
        v = self.d.get(key)
        if v is None:
            return None
        self.d.move_to_end(key)
        return v[0]

    def put(self, key, sp, pinned: bool = False) -> None:
        self.d[key] = (sp, pinned)
        self.d.move_to_end(key)
        for want_pinned, cap in ((False, self.cap), (True, self.cap_pinned)):
            keys = [k for k, (_, pin) in self.d.items() if pin == want_pinned]
            for k in keys[:max(0, len(keys) - cap)]:
                del self.d[k]

    def __setitem__(self, key, sp) -> None:
        self.put(key, sp)

    def __len__(self) -> int:
        return len(self.d)

    def __contains__(self, key) -> bool:
        return key in self.d

    def clear(self) -> None:
        self.d.clear()


class TrunkEngine:
    """Forward/backward of a chain of residual dense blocks, optionally wrapped by the generator's
    head (conv1) and tail (conv2, upsampling, conv3, conv4, clamp).

    Reference: _ResidualDenseBlock.forward BSRGAN/model.py:51-62, _ResidualResidualDenseBlock.forward
    :79-88, BSRGAN._forward_impl :366-381 (RRDBNet: ESRGAN/model.py:208-229)."""

    def __init__(self, owner: nn.Module, rdbs: Sequence[nn.Module], rrdb: bool, full: bool):
        self.owner = owner
        self.rdbs = list(rdbs)
        self.rrdb = rrdb
        self.full = full
        self.R = len(self.rdbs)
        c5 = self.rdbs[0].conv5.weight
        self.Cc = c5.shape[0]                       # channels
        self.G = self.rdbs[0].conv1.weight.shape[0]  # growth channels
        if self.Cc % 32 or self.G % 32:
            raise A.SrganfdError("channels and growth_channels must be multiples of 32 for the MFMA path")
        self.Ccat = self.Cc + 4 * self.G
        if full:
            self.n_up = owner.n_upsample()
            self.in_ch = owner.conv1.weight.shape[1]
            self.out_ch = owner.conv4.weight.shape[0]
        self.fp = FlatParams(list(owner.named_parameters()))
        self.shapes = PlanCache()
        self.packed: Dict[int, dict] = {}
        self.token = 0
        self._rdb_prefix = self._find_prefixes()

    # -- parameter bookkeeping ----------------------------------------------------------------
    def _find_prefixes(self) -> List[str]:
        names = {id(m): n for n, m in self.owner.named_modules()}
        out = []
        for r in self.rdbs:
            n = names[id(r)]
            out.append(n + "." if n else "")
        return out

    def _poff(self, name: str) -> int:
        return self.fp.off(name)

    def _build_pack(self, dtc: int, device) -> dict:
        """Pack-job tables + offsets of every packed operand (forward and data-gradient)."""
        Cc, G, Ccat = self.Cc, self.G, self.Ccat
        jobs, offs, cur = [], {}, 0

        def add(key, ksize, k, n, segs):
            nonlocal cur
            offs[key] = cur
            jobs.append(ops.pack_job(cur, dtc, ksize, k, n, segs))
            cur += (ops.packed_bytes(dtc, ksize, k, n) + 255) // 256 * 256

        def fwd(key, wname, co, ci):
            add(key, 3, ops.pad32(ci), ops.pad32(co), [dict(src_off=self._poff(wname), co_src=co, ci_src=ci, k_len=ops.pad32(ci))])

        def bwd(key, wname, co, ci):
            add(key, 3, ops.pad32(co), ops.pad32(ci), [dict(src_off=self._poff(wname), co_src=co, ci_src=ci, k_len=ops.pad32(co), transposed=1)])

        for i, pre in enumerate(self._rdb_prefix):
            for k in range(1, 6):
                cin, cout = Cc + (k - 1) * G, (Cc if k == 5 else G)
                fwd(("f", i, k), f"{pre}conv{k}.weight", cout, cin)
            # data gradient of the block = dense block over [dOut(C) | dY4 | dY3 | dY2 | dY1]
            s5 = 0.2 * (0.2 if (self.rrdb and i % 3 == 2) else 1.0)
            dyoff = {5: 0, 4: Cc, 3: Cc + G, 2: Cc + 2 * G, 1: Cc + 3 * G}
            for step in range(5):               # step 0..3 -> dY4..dY1, step 4 -> dX
                c_lo = Cc + (3 - step) * G if step < 4 else 0
                n = G if step < 4 else Cc
                kdim = Cc + step * G
                segs = []
                for j in range(5, 4 - step, -1):   # convs whose input covers channels [c_lo, c_lo+n)
                    cin_j, cout_j = Cc + (j - 1) * G, (Cc if j == 5 else G)
                    segs.append(dict(src_off=self._poff(f"{pre}conv{j}.weight"), co_src=cout_j, ci_src=cin_j, k_lo=dyoff[j],
                                     k_len=cout_j, ci_off=c_lo, transposed=1, scale=(s5 if j == 5 else 1.0)))
                add(("b", i, step), 3, kdim, n, segs)
        if self.full:
            fwd(("f", "conv1"), "conv1.weight", Cc, self.in_ch)
            for nm in ["conv2"] + [f"upsampling{u}.0" for u in range(1, self.n_up + 1)] + ["conv3.0"]:
                fwd(("f", nm), nm + ".weight", Cc, Cc)
                bwd(("b", nm), nm + ".weight", Cc, Cc)
            fwd(("f", "conv4"), "conv4.weight", self.out_ch, Cc)
            bwd(("b", "conv4"), "conv4.weight", self.out_ch, Cc)
        table = ops.PackTable(jobs, device)
        buf = torch.empty(cur, dtype=torch.uint8, device=device)
        return dict(table=table, offs=offs, buf=buf)

    def _ensure_packed(self, dtc: int, device) -> dict:
        flat = self.fp.sync(device)
        pk = self.packed.get(dtc)
        if pk is None or pk["buf"].device != device or pk.get("flat_ptr") != flat.data_ptr():
            pk = self._build_pack(dtc, device)
            pk["flat_ptr"] = flat.data_ptr()
            self.packed[dtc] = pk
        if self.fp.stale(pk):
            pk["table"].run(flat, pk["buf"])
        return pk

    # -- per-shape plan -----------------------------------------------------------------------
    def _plan(self, N: int, H: int, W: int, dt: torch.dtype, dtc: int, device, train: bool, pk: dict) -> _Shape:
        # the launch lists bake absolute pointers into the packed buffer AND the flat parameter buffer (biases): both are in the key
        key = (N, H, W, dtc, train, str(device), pk["buf"].data_ptr(), self.fp.flat.data_ptr())
        sp = self.shapes.get(key)
        if sp is not None:
            return sp
        sp = _Shape()
        Cc, G, Ccat, R = self.Cc, self.G, self.Ccat, self.R
        flat = self.fp.flat
        fptr = flat.data_ptr()
        wptr = pk["buf"].data_ptr()

        def new(*shape, dtype=dt):
            return torch.empty(*shape, dtype=dtype, device=device)

        if train:
            cat = sp.cat = [new(N, H, W, Ccat) for _ in range(R + 1)]
            catb = lambda i: cat[i]            # (captures the list, not `sp`: a closure stored on sp that refers to sp is a cycle)
        else:
            # inference: block 0's buffer is kept (conv2 adds out1, model.py:369-370); the rest rotate.
            # block i reads buffer i, writes i+1, and (last block of an RRDB) re-reads i-2: 4 buffers suffice.
            cat = sp.cat = [new(N, H, W, Ccat) for _ in range(min(R + 1, 5))]
            catb = lambda i: cat[0] if i == 0 else cat[1 + (i - 1) % 4]
        sp.catb = catb
        V = A.view
        # the dense-block buffers of the full generator are stored as planar 32-channel groups (srganfd_view.planar): every
        # 32-channel chunk pass of a conv / weight-gradient launch then reads whole contiguous lines instead of 64 of the 384 bytes
        # of each pixel, and the 96- / 160-channel convs stop fetching half-used lines.  Stand-alone blocks keep NHWC (their
        # boundary kernels convert NCHW <-> NHWC directly into / out of these buffers).
        sp.planar = 1 if (self.full and os.environ.get("SRGANFD_PLANAR", "1") != "0") else 0
        VC = lambda t, c0=0: A.view(t, c0=c0, planar=sp.planar)

        def bias(name):
            return fptr + 4 * self._poff(name)

        fw = []   # forward launch list
        if self.full:
            s = 1 << self.n_up
            # thin-side kernels (csrc/conv_thin.hip) for conv1 / conv4 in the 16-bit modes: the image side is NHWC with a 4-channel pitch
            sp.thin_i, sp.thin_o = ops.thin_ok(dtc, Cc, self.in_ch), ops.thin_ok(dtc, Cc, self.out_ch)
            cin1 = ops.pad32(self.in_ch)          # 3 -> 32; Real-ESRGAN below x4: 12 -> 32, 48 -> 64 (Real_ESRGAN/model.py:190-201)
            sp.xin = new(N, H, W, 4 if sp.thin_i else cin1)
            sp.f0 = new(N, H, W, Cc)
            sp.ups = [new(N, H << u, W << u, Cc) for u in range(1, self.n_up + 1)]
            sp.c3 = new(N, H * s, W * s, Cc)
            sp.srp = new(N, H * s, W * s, 4, dtype=torch.float32)
            if sp.thin_i:
                fw.append(ops.ThinLaunch("thin_in", ops.thin_args(dtc, N, H, W, self.in_ch, bias("conv1.weight"), VC(catb(0)), w_big_is_cout=True,
                                                                  bias=bias("conv1.bias"), thin=sp.xin)))
            else:
                fw.append(ops.conv_args(dtc, V(sp.xin), VC(catb(0)), wptr + pk["offs"][("f", "conv1")], N, H, W, cin1, Cc, bias=bias("conv1.bias")))
        for i, pre in enumerate(self._rdb_prefix):
            ci = catb(i)
            blk = []
            for k in range(1, 5):
                blk.append(ops.conv_args(dtc, VC(ci), VC(ci, c0=Cc + (k - 1) * G), wptr + pk["offs"][("f", i, k)], N, H, W, Cc + (k - 1) * G, G,
                                         bias=bias(f"{pre}conv{k}.bias"), act=A.ACT_LRELU, slope=0.2))
            last = self.rrdb and i % 3 == 2
            kw = dict(post_scale=0.04, r1=VC(ci), r1_scale=0.2, r2=VC(catb(i - 2)), r2_scale=1.0) if last else \
                dict(post_scale=0.2, r1=VC(ci), r1_scale=1.0)
            blk.append(ops.conv_args(dtc, VC(ci), VC(catb(i + 1)), wptr + pk["offs"][("f", i, 5)], N, H, W, Ccat, Cc,
                                     bias=bias(f"{pre}conv5.bias"), **kw))
            # small batches (the reference's own crop sizes): the five launches as ONE LDS-resident launch (csrc/dense_chain.hip)
            fw.extend(ops.dense_chain_or_launches(blk, device) if (Cc, G) == (64, 32) else blk)
        if self.full:
            tout = catb(R)
            fw.append(ops.conv_args(dtc, VC(tout), V(sp.f0), wptr + pk["offs"][("f", "conv2")], N, H, W, Cc, Cc, bias=bias("conv2.bias"),
                                    r1=VC(catb(0)), r1_scale=1.0))
            src, h, w = sp.f0, H, W
            for u in range(1, self.n_up + 1):
                nm = f"upsampling{u}.0"
                fw.append(ops.conv_args(dtc, V(src), V(sp.ups[u - 1]), wptr + pk["offs"][("f", nm)], N, h, w, Cc, Cc, up=1,
                                        bias=bias(nm + ".bias"), act=A.ACT_LRELU, slope=0.2))
                src, h, w = sp.ups[u - 1], h * 2, w * 2
            fw.append(ops.conv_args(dtc, V(src), V(sp.c3), wptr + pk["offs"][("f", "conv3.0")], N, h, w, Cc, Cc, bias=bias("conv3.0.bias"),
                                    act=A.ACT_LRELU, slope=0.2))
            if sp.thin_o:
                fw.append(ops.ThinLaunch("thin_out", ops.thin_args(dtc, N, h, w, self.out_ch, bias("conv4.weight"), V(sp.c3), w_big_is_cout=False,
                                                                   bias=bias("conv4.bias"), thin_out=sp.srp, thin_out_pitch=4)))
            else:
                fw.append(ops.conv_args(dtc, V(sp.c3), V(sp.srp), wptr + pk["offs"][("f", "conv4")], N, h, w, Cc, 32, cout_store=self.out_ch,
                                        bias=bias("conv4.bias"), y_f32=True))
            sp.hs, sp.ws = h, w
        sp.fw = fw
        sp.N, sp.H, sp.W, sp.dt, sp.dtc, sp.device = N, H, W, dt, dtc, device
        if train:
            self._plan_backward(sp, pk)
        self.shapes.put(key, sp, pinned=train)
        return sp

    def _plan_backward(self, sp: _Shape, pk: dict) -> None:
        N, H, W, dt, dtc, device = sp.N, sp.H, sp.W, sp.dt, sp.dtc, sp.device
        Cc, G, Ccat, R = self.Cc, self.G, self.Ccat, self.R
        wptr = pk["buf"].data_ptr()
        V = A.view
        VC = lambda t, c0=0: A.view(t, c0=c0, planar=sp.planar)      # forward dense-block buffers
        VD = VC                                                       # stacked-gradient buffers (same layout: planar only one of
        #                                                               the two measured half the gain each)

        def new(*shape, dtype=dt):
            return torch.empty(*shape, dtype=dtype, device=device)

        sp.dy = [new(N, H, W, Ccat) for _ in range(4)]
        sp.dx0 = new(N, H, W, Cc)           # gradient w.r.t. the trunk input
        dyb = lambda i: sp.dy[i % 4]
        bw: List[tuple] = []                 # ("conv", args) | ("wgrad", plan, xview, dyview, grad_off) | ("call", fn)
        ws_bytes = 0

        def wplan(n, h, w, xch, dych, convs, up=0):
            nonlocal ws_bytes
            p = ops.WgradPlan(device, dtc, n, h, w, xch, dych, convs, up=up)
            ws_bytes = max(ws_bytes, p.workspace_bytes)
            return p

        if self.full:
            s = 1 << self.n_up
            hs, ws_ = sp.hs, sp.ws
            sp.dsrp = new(N, hs, ws_, 4 if sp.thin_o else 32)
            sp.gA = new(N, hs, ws_, Cc)
            fptr = self.fp.flat.data_ptr()
            if sp.thin_i or sp.thin_o:
                sp.thin_ws = torch.empty(ops.thin_wgrad_workspace_bytes(), dtype=torch.uint8, device=device)
            sp.gB = new(N, hs, ws_, Cc)
            sp.glo = [new(N, H << u, W << u, Cc) for u in range(0, self.n_up)]   # grads at the input res of upsampling u+1

            def one(name, cout, cin, dw_only=False):
                return [dict(cin=ops.pad32(cin), cout=ops.pad32(cout), dw_off=self._poff(name + ".weight"), db_off=self._poff(name + ".bias"),
                             co_dst=cout, ci_dst=cin)]
            # conv4
            if sp.thin_o:
                w4 = fptr + 4 * self._poff("conv4.weight")
                bw.append(("thin", ops.ThinLaunch("thin_wgrad", ops.thin_args(dtc, N, hs, ws_, self.out_ch, w4, V(sp.c3), w_big_is_cout=False, thin=sp.dsrp),
                                                  dw_off=self._poff("conv4.weight"), db_off=self._poff("conv4.bias"), ws=sp.thin_ws)))
                bw.append(("thin", ops.ThinLaunch("thin_in", ops.thin_args(dtc, N, hs, ws_, self.out_ch, w4, V(sp.gA), w_big_is_cout=False, flip=True,
                                                                           mask=V(sp.c3), mask_slope=0.2, thin=sp.dsrp))))
            else:
                bw.append(("wgrad", wplan(N, hs, ws_, Cc, 32, one("conv4", self.out_ch, Cc)), V(sp.c3), V(sp.dsrp), 0))
                bw.append(("conv", ops.conv_args(dtc, V(sp.dsrp), V(sp.gA), wptr + pk["offs"][("b", "conv4")], N, hs, ws_, 32, Cc, mask=V(sp.c3), mask_slope=0.2)))
            # conv3
            src3 = sp.ups[-1] if self.n_up else sp.f0
            bw.append(("wgrad", wplan(N, hs, ws_, Cc, Cc, one("conv3.0", Cc, Cc)), V(src3), V(sp.gA), 0))
            if self.n_up:
                bw.append(("conv", ops.conv_args(dtc, V(sp.gA), V(sp.gB), wptr + pk["offs"][("b", "conv3.0")], N, hs, ws_, Cc, Cc, mask=V(src3), mask_slope=0.2)))
            else:
                bw.append(("conv", ops.conv_args(dtc, V(sp.gA), V(sp.gB), wptr + pk["offs"][("b", "conv3.0")], N, hs, ws_, Cc, Cc)))
            cur, other = sp.gB, sp.gA        # cur = gradient w.r.t. pre-activation of upsampling{n_up} (at its output res)
            for u in range(self.n_up, 0, -1):
                nm = f"upsampling{u}.0"
                hin, win = H << (u - 1), W << (u - 1)
                xin_t = sp.ups[u - 2] if u >= 2 else sp.f0
                npx = N * (hin * 2) * (win * 2)
                cur_v = V(cur.view(-1)[: npx * Cc].view(N, hin * 2, win * 2, Cc))
                oth_v = V(other.view(-1)[: npx * Cc].view(N, hin * 2, win * 2, Cc))
                bw.append(("wgrad", wplan(N, hin, win, Cc, Cc, one(nm, Cc, Cc), up=1), V(xin_t), cur_v, 0))
                bw.append(("conv", ops.conv_args(dtc, cur_v, oth_v, wptr + pk["offs"][("b", nm)], N, hin * 2, win * 2, Cc, Cc)))
                glo = sp.glo[u - 1]
                bw.append(("call", (lambda a=oth_v, b=V(glo), hh=hin, ww=win: A.check(
                    A.lib().srganfd_resample(0, a, b, dtc, N, hh, ww, Cc, A.stream_ptr()), "nearest_bwd"))))
                if u >= 2:   # input of this stage is the LeakyReLU output of the previous upsampling conv
                    bw.append(("call", (lambda d=V(glo), act=V(xin_t), npx2=N * hin * win: A.check(
                        A.lib().srganfd_lrelu_bwd(d, act, A.NULL_VIEW, d, dtc, npx2, Cc, 0.2, A.stream_ptr()), "lrelu_bwd"))))
                    # next stage works at (hin, win): reuse gA/gB as scratch, the gradient lives in glo
                    cur, other = glo, sp.gA
                    # make cur the full buffer view expected above
                else:
                    cur = glo
            d_f0 = cur if self.n_up else sp.gB
            sp.d_f0 = d_f0
            # conv2: f0 = out1 + conv2(trunk_out)
            bw.append(("wgrad", wplan(N, H, W, Cc, Cc, one("conv2", Cc, Cc)), VC(sp.catb(R)), V(d_f0), 0))
            bw.append(("conv", ops.conv_args(dtc, V(d_f0), VD(dyb(R - 1)), wptr + pk["offs"][("b", "conv2")], N, H, W, Cc, Cc)))
            # gradient buckets for the data-parallel exchange (parallel.BucketReducer): parameters sit in named_parameters() order --
            # conv1, the dense blocks, conv2 and the tail -- so "everything from conv2 on" is one contiguous range, final here
            bw.append(("ready", self._poff("conv2.weight"), self.fp.total))
        # dense blocks, last to first
        rdb_names = ["conv%d" % k for k in range(1, 6)]
        plans = {}
        for i in range(R - 1, -1, -1):
            pre = self._rdb_prefix[i]
            last = self.rrdb and i % 3 == 2
            first = self.rrdb and i % 3 == 0
            s_out = 0.2 if last else 1.0
            s5 = 0.2 * s_out
            ci, di = sp.catb(i), dyb(i)
            if s5 not in plans:
                base = self._poff(pre + "conv1.weight")
                convs = []
                for k in range(1, 6):
                    cin, cout = Cc + (k - 1) * G, (Cc if k == 5 else G)
                    convs.append(dict(ci_lo=0, cin=cin, co_lo=(0 if k == 5 else Cc + (4 - k) * G), cout=cout,
                                      dw_off=self._poff(pre + f"conv{k}.weight") - base, db_off=self._poff(pre + f"conv{k}.bias") - base,
                                      co_dst=cout, ci_dst=cin, alpha=(s5 if k == 5 else 1.0)))
                plans[s5] = wplan(N, H, W, Ccat, Ccat, convs)
            blk = []
            for step in range(4):
                kdim = Cc + step * G
                blk.append(ops.conv_args(dtc, VD(di), VD(di, c0=Cc + step * G), wptr + pk["offs"][("b", i, step)], N, H, W, kdim, G,
                                         mask=VC(ci, c0=Cc + (3 - step) * G), mask_slope=0.2))
            dst = VD(dyb(i - 1)) if i > 0 else V(sp.dx0)
            kw = dict(r1=VD(di), r1_scale=s_out)
            if first:
                kw.update(r2=VD(dyb(i + 2)), r2_scale=1.0)
            blk.append(ops.conv_args(dtc, VD(di), dst, wptr + pk["offs"][("b", i, 4)], N, H, W, Ccat, Cc, **kw))
            chain = ops.dense_chain_or_launches(blk, sp.device) if (Cc, G) == (64, 32) else blk
            wg = ("wgrad", plans[s5], VC(ci), VD(di), self._poff(pre + "conv1.weight"), i)     # 6th field: dense block (batched reduction)
            fence = [("fence", i + 3)] if i > 0 else []     # dst is the buffer block i + 3's weight gradient read (side-stream mode, see backward())
            ready = [("ready", self._poff(pre + "conv1.weight"), self._poff("conv2.weight"))] if self.full and R >= 4 and i == R // 2 else []     # upper half of the trunk is final: second bucket
            if len(chain) == 1:
                # one launch for the five data-gradient convs (it writes dst, so the fence comes first); the weight gradient reads what
                # the chain's four growth layers wrote and follows it
                bw.extend(fence + [("chain", chain[0]), wg] + ready)
            else:
                bw.extend([("conv", a_) for a_ in blk[:4]] + [wg] + ready + fence + [("conv", blk[4])])
        if self.full:
            bw.append(("call", (lambda x=V(sp.d_f0), y=V(sp.dx0): A.check(
                A.lib().srganfd_axpby(x, y, dtc, N * H * W, Cc, 1.0, 1.0, A.stream_ptr()), "axpby"))))
            if sp.thin_i:
                bw.append(("thin", ops.ThinLaunch("thin_wgrad", ops.thin_args(dtc, N, H, W, self.in_ch, self.fp.flat.data_ptr() + 4 * self._poff("conv1.weight"),
                                                                              V(sp.dx0), w_big_is_cout=True, thin=sp.xin),
                                                  dw_off=self._poff("conv1.weight"), db_off=self._poff("conv1.bias"), ws=sp.thin_ws)))
            else:
                cin1 = ops.pad32(self.in_ch)
                convs = [dict(cin=cin1, cout=Cc, dw_off=self._poff("conv1.weight"), db_off=self._poff("conv1.bias"), co_dst=Cc, ci_dst=self.in_ch)]
                bw.append(("wgrad", wplan(N, H, W, cin1, Cc, convs), V(sp.xin), V(sp.dx0), 0))
        # last bucket: whatever the earlier markers did not cover
        covered = min([it[1] for it in bw if it[0] == "ready"], default=self.fp.total)
        bw.append(("ready", 0, covered))
        sp.bw = bw
        sp.wg_ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
        # four more workspaces of the dense-block plan's size for the batched slab reduction (34 MB each at B=32, 128x128)
        dense_ws = max((p_.workspace_bytes for p_ in plans.values()), default=0)
        sp.wg_ws4 = [torch.empty(dense_ws, dtype=torch.uint8, device=device) for _ in range(_BATCH_REDUCE)] if (dense_ws and _BATCH_REDUCE > 1) else None

    # -- execution ----------------------------------------------------------------------------
    def forward(self, x: Tensor, train: bool) -> Tensor:
        """x: NCHW fp32 (generator: image; trunk-only: feature map).  Returns NCHW fp32."""
        _require_gpu(x)
        dt, dtc = _dt(self.owner)
        dev = x.device
        pk = self._ensure_packed(dtc, dev)
        N, _, H, W = x.shape
        sp = self._plan(N, H, W, dt, dtc, dev, train, pk)
        x = x.contiguous().float()
        L, st = A.lib(), A.stream_ptr()
        if self.full:
            A.check(L.srganfd_nchw_to_nhwc(x.data_ptr(), N, self.in_ch, H, W, A.view(sp.xin), dtc, sp.xin.shape[-1], None, None, st), "nchw_to_nhwc")
        else:
            A.check(L.srganfd_nchw_to_nhwc(x.data_ptr(), N, self.Cc, H, W, A.view(sp.catb(0)), dtc, self.Cc, None, None, st), "nchw_to_nhwc")
        rec = profiling.REC
        Thin, Chain = ops.ThinLaunch, ops.DenseChain
        if rec is None:
            for a in sp.fw:
                if type(a) is Thin or type(a) is Chain:
                    a.run()
                    continue
                rc = L.srganfd_conv2d(C.byref(a), st)
                if rc:
                    A.check(rc, "conv2d")
        else:
            for a in sp.fw:
                if type(a) is Thin or type(a) is Chain:
                    a.launch(rec)
                    continue
                rec.bracket(profiling.conv_label(a), profiling.conv_work(a), lambda: A.check(L.srganfd_conv2d(C.byref(a), st), "conv2d"))
        if self.full:
            out = torch.empty(N, self.out_ch, sp.hs, sp.ws, dtype=torch.float32, device=dev)
            A.check(L.srganfd_nhwc_to_nchw(A.view(sp.srp), A.F32, N, self.out_ch, sp.hs, sp.ws, out.data_ptr(), 1, st), "nhwc_to_nchw")
        else:
            out = torch.empty(N, self.Cc, H, W, dtype=torch.float32, device=dev)
            A.check(L.srganfd_nhwc_to_nchw(A.view(sp.catb(self.R)), dtc, N, self.Cc, H, W, out.data_ptr(), 0, st), "nhwc_to_nchw")
        if train:
            self.token += 1
            sp.token = self.token
        self._last = sp
        return out

    def backward(self, sp: _Shape, token: int, dout: Tensor, need_dx: bool, on_ready=None) -> Tuple[Tensor, Optional[Tensor]]:
        """dout: NCHW fp32 gradient of forward()'s result.  Returns (flat parameter gradient, dx or None).
        ``on_ready(flat_grad, lo, hi)`` (data parallelism, parallel.BucketReducer.bucket) is called when every launch that writes
        elements [lo, hi) of the flat gradient has been enqueued; the ranges are disjoint and cover the whole buffer."""
        if getattr(sp, "token", None) != token:
            raise A.SrganfdError("generator activations were overwritten by a later training-mode forward before backward ran")
        L, st = A.lib(), A.stream_ptr()
        N, H, W, dtc = sp.N, sp.H, sp.W, sp.dtc
        dout = dout.contiguous().float()
        flat_grad = self.fp.new_grad(sp.device)
        if self.full:
            A.check(L.srganfd_clamp_grad_to_nhwc(dout.data_ptr(), A.view(sp.srp), N, self.out_ch, sp.hs, sp.ws, A.view(sp.dsrp), dtc, sp.dsrp.shape[-1], st), "clamp_grad")
        else:
            A.check(L.srganfd_nchw_to_nhwc(dout.data_ptr(), N, self.Cc, H, W, A.view(sp.dy[(self.R - 1) % 4]), dtc, self.Cc, None, None, st), "nchw_to_nhwc")
        gptr = flat_grad.data_ptr()
        rec = profiling.REC
        pend_red = []          # dense-block weight-gradient launches whose slabs wait for the batched reduction
        # SRGANFD_WGRAD_STREAM=1: the dense blocks' weight-gradient launches (and their slab reductions) run on a second stream beside
        # the data-gradient chain of the following blocks -- they only read what the chain has finished (block i's stacked gradient and
        # activations); the chain waits ("fence") before it overwrites a gradient buffer a pending weight gradient still reads.
        side = None
        if _WGRAD_STREAM and sp.wg_ws4 is not None:
            side = getattr(sp, "wg_stream", None)
            if side is None:
                side = sp.wg_stream = torch.cuda.Stream(device=sp.device)
            main = torch.cuda.current_stream()
            wg_done = {}       # dense block -> event behind its weight-gradient launch on the side stream

        def on_side(fn):
            if side is None:
                return fn()
            with torch.cuda.stream(side):
                return fn()

        def flush_reduce():
            if not pend_red:
                return
            jobs = (A.WgradReduceJob * len(pend_red))()
            for j, (plan, goff, ws) in zip(jobs, pend_red):
                j.plan_host, j.plan_dev = C.addressof(plan.host), plan.dev.data_ptr()
                j.grads, j.scalars, j.workspace = gptr + 4 * goff, None, ws.data_ptr()
            n_jobs = len(pend_red)
            run = lambda: A.check(L.srganfd_wgrad_reduce_batch(jobs, n_jobs, A.stream_ptr()), "wgrad_reduce_batch")
            if rec is None:
                on_side(run)
            else:
                on_side(lambda: rec.bracket("wgrad_reduce_batch", (0.0, float(sum(w.numel() for _, _, w in pend_red))), run))
            pend_red.clear()
        for item in sp.bw:
            kind = item[0]
            if kind == "conv":
                if rec is None:
                    rc = L.srganfd_conv2d(C.byref(item[1]), st)
                    if rc:
                        A.check(rc, "conv2d(dgrad)")
                else:
                    a = item[1]
                    rec.bracket(profiling.conv_label(a), profiling.conv_work(a), lambda: A.check(L.srganfd_conv2d(C.byref(a), st), "conv2d(dgrad)"))
            elif kind == "chain":
                item[1].launch(rec)
            elif kind == "wgrad":
                plan, xv, dyv, goff = item[1:5]
                if len(item) > 5 and sp.wg_ws4 is not None:
                    # dense block: MFMA kernel now, slabs into one of four workspaces; the slab reduction of up to four blocks is ONE
                    # launch (srganfd_wgrad_reduce_batch: the reduction is latency-bound at ~20 us whatever it reduces)
                    ws = sp.wg_ws4[len(pend_red)]
                    run = lambda: A.check(L.srganfd_conv2d_wgrad_partial(plan.host, plan.dev.data_ptr(), xv, dyv, ws.data_ptr(), ws.numel(), A.stream_ptr()),
                                          "conv2d_wgrad_partial")
                    if side is not None:
                        side.wait_stream(main)          # the block's four data-gradient launches have written its stacked gradient
                    if rec is None:
                        on_side(run)
                    else:
                        on_side(lambda: rec.bracket(plan.label, (plan.flops, plan.nbytes), run))
                    if side is not None:
                        ev = torch.cuda.Event()
                        ev.record(side)
                        wg_done[item[5]] = ev
                    pend_red.append((plan, goff, ws))
                    if len(pend_red) == len(sp.wg_ws4):
                        flush_reduce()
                else:
                    flush_reduce()                      # the launches share workspace slot 0
                    run = lambda: A.check(L.srganfd_conv2d_wgrad(plan.host, plan.dev.data_ptr(), xv, dyv, gptr + 4 * goff, None,
                                                                 sp.wg_ws.data_ptr(), sp.wg_ws.numel(), st), "conv2d_wgrad")
                    if rec is None:
                        run()
                    else:
                        rec.bracket(plan.label, (plan.flops, plan.nbytes), run)
            elif kind == "thin":
                flush_reduce()                          # (keeps the launch order of the padded path: slot-0 workspace users come after pending slabs)
                item[1].launch(rec, gptr)
            elif kind == "ready":
                flush_reduce()
                if side is not None:
                    main.wait_stream(side)
                if on_ready is not None:
                    on_ready(flat_grad, item[1], item[2])
            elif kind == "fence":
                if side is not None and item[1] in wg_done:
                    main.wait_event(wg_done.pop(item[1]))
            else:
                item[1]()
        flush_reduce()
        if side is not None:
            main.wait_stream(side)
        dx = None
        if need_dx and not self.full:
            dx = torch.empty(N, self.Cc, H, W, dtype=torch.float32, device=sp.device)
            A.check(L.srganfd_nhwc_to_nchw(A.view(sp.dx0), dtc, N, self.Cc, H, W, dx.data_ptr(), 0, st), "nhwc_to_nchw")
        return flat_grad, dx


# dense blocks whose weight-gradient slabs one srganfd_wgrad_reduce_batch launch reduces (<= 8 = SRGANFD's kRedBatch; 0 / 1: every block's own)
_BATCH_REDUCE = max(0, min(8, int(os.environ.get("SRGANFD_BATCH_REDUCE", "4"))))
_WGRAD_STREAM = int(os.environ.get("SRGANFD_WGRAD_STREAM", "0"))


class _TrunkFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eng, train, *params):
        out = eng.forward(x, train)
        ctx.eng, ctx.sp, ctx.token, ctx.train = eng, eng._last, eng.token, train
        ctx.need_dx = ctx.needs_input_grad[0]
        return out

    @staticmethod
    def backward(ctx, dout):
        if not ctx.train:
            raise A.SrganfdError("backward through a forward that ran without gradients")
        flat_grad, dx = ctx.eng.backward(ctx.sp, ctx.token, dout, ctx.need_dx)
        return (dx, None, None) + tuple(ctx.eng.fp.grad_views(flat_grad))


def _engine(owner: nn.Module, factory):
    eng = _ENGINES.get(owner)
    if eng is None:
        eng = factory()
        # the registry is keyed weakly by the module; the engine must not keep the module alive in turn, or the entry --
        # and every activation buffer of the engine -- would never be released (found at 286 GB by test_fullsize_gpu.py)
        if getattr(eng, "owner", None) is owner:
            eng.owner = weakref.proxy(owner)
        if hasattr(eng, "rdbs"):                      # a stand-alone dense block is its own engine's only block
            eng.rdbs = [weakref.proxy(m) if m is owner else m for m in eng.rdbs]
        _ENGINES[owner] = eng
    return eng


def _run_trunk(eng: "TrunkEngine", x: Tensor) -> Tensor:
    train = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in eng.fp.params))
    if not train:
        return eng.forward(x, False)
    return _TrunkFn.apply(x, eng, True, *eng.fp.params)


def trunk_apply(owner: nn.Module, x: Tensor, rdbs: Sequence[nn.Module], rrdb: bool) -> Tensor:
    return _run_trunk(_engine(owner, lambda: TrunkEngine(owner, rdbs, rrdb, full=False)), x)


def generator_engine(owner: nn.Module) -> TrunkEngine:
    def make():
        rdbs = []
        for blk in owner.trunk:
            rdbs += [blk.rdb1, blk.rdb2, blk.rdb3]
        return TrunkEngine(owner, rdbs, rrdb=True, full=True)
    return _engine(owner, make)


def generator_apply(owner: nn.Module, x: Tensor) -> Tensor:
    return _run_trunk(generator_engine(owner), x)


def discriminator_apply(owner: nn.Module, x: Tensor) -> Tensor:
    from .engine_d import discriminator_apply as f
    return f(owner, x)


def content_loss_apply(owner: nn.Module, sr: Tensor, gt: Tensor) -> Tensor:
    from .engine_d import content_loss_apply as f
    return f(owner, sr, gt)
