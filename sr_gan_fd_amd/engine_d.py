"""HIP engines for the U-Net discriminator (spectral-normalised) and the VGG-19 content loss.

Reference: DiscriminatorUNet._forward_impl BSRGAN/model.py:141-167 with torch's spectral_norm
(torch/nn/utils/spectral_norm.py:62-114) applied at :104-132; ContentLoss.forward :536-554.

Discriminator data flow (NHWC, one buffer per saved activation):
  x(3->32 pad) -conv1-> out1 -4x4s2-> d1 -4x4s2-> d2 -4x4s2-> d3 -bilinear-> b3 -conv-> (+d2) u1
  -bilinear-> b2 -conv-> (+d1) u2 -bilinear-> b1 -conv-> (+out1) u3 -conv-> c2 -conv-> c3 -conv4-> logits(fp32)
Spectral norm: per training forward one power iteration (HIP kernels) updates weight_u / weight_v in
place and produces 1/sigma on the device; the weight packer multiplies it in, so the conv kernels see
W/sigma without an extra pass.  Backward: dL/d(W/sigma) from the wgrad kernel goes through
srganfd_spectral_norm_grad (both the 1/sigma path and the -<G,W>/sigma^2 u v^T path).
The stride-2 data gradient runs as 4 output-parity classes, each a 2x2-tap stride-1 conv.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import torch
from torch import Tensor, nn

from . import _abi as A
from . import ops
from . import profiling
from .engine import FlatParams, _ENGINES, _dt, _engine, _require_gpu, _Shape, PlanCache

SN_LAYERS = [("down_block1", 4, 2), ("down_block2", 4, 2), ("down_block3", 4, 2), ("up_block1", 3, 1),
             ("up_block2", 3, 1), ("up_block3", 3, 1), ("conv2", 3, 1), ("conv3", 3, 1)]


class DiscriminatorEngine:
    def __init__(self, owner: nn.Module):
        self.owner = owner
        self.fp = FlatParams(list(owner.named_parameters()))
        self.in_ch = owner.conv1.weight.shape[1]
        self.out_ch = owner.conv4.weight.shape[0]
        ch = owner.down_block1[0].weight_orig.shape[1]
        if ch != 64 or owner.conv1.weight.shape[0] != 64:
            raise A.SrganfdError("DiscriminatorUNet: conv1 hard-codes 64 outputs (model.py:102), so channels must be 64")
        self.ch = ch
        self.dims = {}   # layer -> (cout, cin, k, stride)
        for name, k, s in SN_LAYERS:
            w = getattr(owner, name)[0].weight_orig
            self.dims[name] = (w.shape[0], w.shape[1], k, s)
        self.shapes = PlanCache()
        self.packed: Dict[int, dict] = {}
        self.token = 0

    def _poff(self, n):
        return self.fp.off(n)

    # ---- packing: every operand carries 1/sigma of its layer (scalars[2*l+1]) ----
    def _build_pack(self, dtc: int, device) -> dict:
        jobs, offs, cur = [], {}, 0

        def add(key, ksize, k, n, seg):
            nonlocal cur
            offs[key] = cur
            jobs.append(ops.pack_job(cur, dtc, ksize, k, n, [seg]))
            cur += (ops.packed_bytes(dtc, ksize, k, n) + 255) // 256 * 256

        add(("f", "conv1"), 3, 32, 64, dict(src_off=self._poff("conv1.weight"), co_src=64, ci_src=self.in_ch, k_len=32))
        add(("b", "conv1"), 3, 64, 32, dict(src_off=self._poff("conv1.weight"), co_src=64, ci_src=self.in_ch, k_len=64, transposed=1))
        add(("f", "conv4"), 3, 64, 32, dict(src_off=self._poff("conv4.weight"), co_src=self.out_ch, ci_src=64, k_len=64))
        add(("b", "conv4"), 3, 32, 64, dict(src_off=self._poff("conv4.weight"), co_src=self.out_ch, ci_src=64, k_len=32, transposed=1))
        for l, (name, k, s) in enumerate(SN_LAYERS):
            co, ci, _, _ = self.dims[name]
            src = self._poff(f"{name}.0.weight_orig")
            add(("f", name), k, ci, co, dict(src_off=src, co_src=co, ci_src=ci, k_len=ci, scale_off=2 * l + 1))
            if s == 1:
                add(("b", name), k, co, ci, dict(src_off=src, co_src=co, ci_src=ci, k_len=co, transposed=1, scale_off=2 * l + 1))
            else:
                for par in range(4):
                    add(("b", name, par), 2, co, ci, dict(src_off=src, co_src=co, ci_src=ci, k_len=co, transposed=2 + par, scale_off=2 * l + 1))
        return dict(table=ops.PackTable(jobs, device), offs=offs, buf=torch.empty(cur, dtype=torch.uint8, device=device))

    def _ensure_packed(self, dtc: int, device) -> dict:
        flat = self.fp.sync(device)
        pk = self.packed.get(dtc)
        if pk is None or pk["buf"].device != device or pk.get("flat_ptr") != flat.data_ptr():
            pk = self._build_pack(dtc, device)
            pk["flat_ptr"] = flat.data_ptr()
            pk["scalars"] = torch.ones(2 * len(SN_LAYERS), dtype=torch.float32, device=device)
            pk["sn_ws"] = torch.empty(sum(A.sn_ws_floats(self.dims[n][0], self.dims[n][1] * k * k) for n, k, _ in SN_LAYERS),
                                      dtype=torch.float32, device=device)
            self.packed[dtc] = pk
        return pk

    def _spectral_norm_and_pack(self, pk: dict, training: bool) -> None:
        flat = self.fp.flat
        sc = pk["scalars"].data_ptr()
        layers = []
        for l, (name, k, s) in enumerate(SN_LAYERS):
            co, ci, _, _ = self.dims[name]
            m = getattr(self.owner, name)[0]
            u, v = m.weight_u, m.weight_v
            if u.device != flat.device or not u.is_contiguous() or not v.is_contiguous():
                raise A.SrganfdError("spectral-norm buffers must live on the module's GPU")
            layers.append((flat.data_ptr() + 4 * self._poff(f"{name}.0.weight_orig"), u.data_ptr(), v.data_ptr(), co, ci * k * k,
                           sc + 8 * l, sc + 8 * l + 4))
        ops.spectral_norm_batch(layers, training, pk["sn_ws"])            # all eight layers in four launches
        pk["table"].run(flat, pk["buf"], pk["scalars"])

    # ---- per-shape plan ----
    def _plan(self, N, S1, S2, dt, dtc, device, pk) -> _Shape:
        key = (N, S1, S2, dtc, str(device), pk["buf"].data_ptr(), self.fp.flat.data_ptr())
        sp = self.shapes.get(key)
        if sp is not None:
            return sp
        if S1 % 8 or S2 % 8:
            raise A.SrganfdError("DiscriminatorUNet input height/width must be multiples of 8")
        sp = _Shape()
        sp.N, sp.H, sp.W, sp.dt, sp.dtc, sp.device = N, S1, S2, dt, dtc, device
        V = A.view
        fptr, wptr = self.fp.flat.data_ptr(), pk["buf"].data_ptr()
        O = pk["offs"]

        def new(h, w, c, dtype=dt):
            return torch.empty(N, h, w, c, dtype=dtype, device=device)
        H, W = S1, S2
        # conv1 (in_ch -> 64) and conv4 (64 -> out_ch) on the thin-side kernels in the 16-bit modes (csrc/conv_thin.hip): 4-channel pitch
        sp.thin_i, sp.thin_o = ops.thin_ok(dtc, 64, self.in_ch), ops.thin_ok(dtc, 64, self.out_ch)
        sp.xin = new(H, W, 4 if sp.thin_i else 32)
        sp.out1 = new(H, W, 64)
        sp.d1, sp.d2, sp.d3 = new(H // 2, W // 2, 128), new(H // 4, W // 4, 256), new(H // 8, W // 8, 512)
        sp.b3, sp.u1 = new(H // 4, W // 4, 512), new(H // 4, W // 4, 256)
        sp.b2, sp.u2 = new(H // 2, W // 2, 256), new(H // 2, W // 2, 128)
        sp.b1, sp.u3 = new(H, W, 128), new(H, W, 64)
        sp.c2, sp.c3 = new(H, W, 64), new(H, W, 64)
        # LeakyReLU outputs before the skip adds (exact derivative sign in backward)
        sp.a1, sp.a2, sp.a3 = new(H // 4, W // 4, 256), new(H // 2, W // 2, 128), new(H, W, 64)
        L = A.lib()
        cv = lambda *a, **k: ("conv", ops.conv_args(dtc, *a, **k))
        rs = lambda op, a, b, h, w, c: ("call", lambda: A.check(L.srganfd_resample(op, a, b, dtc, N, h, w, c, A.stream_ptr()), "resample"))
        lre = dict(act=A.ACT_LRELU, slope=0.2)
        w1, b1 = fptr + 4 * self._poff("conv1.weight"), fptr + 4 * self._poff("conv1.bias")
        fw = [
            ("thin", ops.ThinLaunch("thin_in", ops.thin_args(dtc, N, H, W, self.in_ch, w1, V(sp.out1), w_big_is_cout=True, bias=b1, thin=sp.xin)))
            if sp.thin_i else cv(V(sp.xin), V(sp.out1), wptr + O[("f", "conv1")], N, H, W, 32, 64, bias=b1),
            cv(V(sp.out1), V(sp.d1), wptr + O[("f", "down_block1")], N, H, W, 64, 128, ksize=4, stride=2, **lre),
            cv(V(sp.d1), V(sp.d2), wptr + O[("f", "down_block2")], N, H // 2, W // 2, 128, 256, ksize=4, stride=2, **lre),
            cv(V(sp.d2), V(sp.d3), wptr + O[("f", "down_block3")], N, H // 4, W // 4, 256, 512, ksize=4, stride=2, **lre),
            rs(1, V(sp.d3), V(sp.b3), H // 8, W // 8, 512),
            cv(V(sp.b3), V(sp.u1), wptr + O[("f", "up_block1")], N, H // 4, W // 4, 512, 256, r1=V(sp.d2), r1_scale=1.0, y2=V(sp.a1), **lre),
            rs(1, V(sp.u1), V(sp.b2), H // 4, W // 4, 256),
            cv(V(sp.b2), V(sp.u2), wptr + O[("f", "up_block2")], N, H // 2, W // 2, 256, 128, r1=V(sp.d1), r1_scale=1.0, y2=V(sp.a2), **lre),
            rs(1, V(sp.u2), V(sp.b1), H // 2, W // 2, 128),
            cv(V(sp.b1), V(sp.u3), wptr + O[("f", "up_block3")], N, H, W, 128, 64, r1=V(sp.out1), r1_scale=1.0, y2=V(sp.a3), **lre),
            cv(V(sp.u3), V(sp.c2), wptr + O[("f", "conv2")], N, H, W, 64, 64, **lre),
            cv(V(sp.c2), V(sp.c3), wptr + O[("f", "conv3")], N, H, W, 64, 64, **lre),
        ]
        sp.fw = fw
        # locals only: a closure stored on sp that captured `sp` or `self` would be a reference cycle, and the plan's activation
        # buffers would then outlive the module until a cyclic collection (found as an OOM between full-size tests)
        c3_v, out_ch, w4, b4 = V(sp.c3), self.out_ch, wptr + O[("f", "conv4")], fptr + 4 * self._poff("conv4.bias")
        if sp.thin_o:
            w4raw = fptr + 4 * self._poff("conv4.weight")
            sp.conv4 = lambda logits: ("thin", ops.ThinLaunch("thin_out", ops.thin_args(dtc, N, H, W, out_ch, w4raw, c3_v, w_big_is_cout=False, bias=b4,
                                                                                        thin_out=logits.data_ptr(), thin_out_pitch=1)))
        else:
            sp.conv4 = lambda logits: ("conv", ops.conv_args(dtc, c3_v, A.View(logits.data_ptr(), out_ch, 0), w4, N, H, W, 64, 32,
                                                             cout_store=out_ch, bias=b4, y_f32=True))
        self._plan_backward(sp, pk)
        self.shapes[key] = sp
        return sp

    def _plan_backward(self, sp: _Shape, pk: dict) -> None:
        N, H, W, dt, dtc, device = sp.N, sp.H, sp.W, sp.dt, sp.dtc, sp.device
        V = A.view
        wptr, O = pk["buf"].data_ptr(), pk["offs"]
        L = A.lib()

        def new(h, w, c, dtype=dt):
            return torch.empty(N, h, w, c, dtype=dtype, device=device)
        sp.dl = new(H, W, 4 if sp.thin_o else 32)
        fptr = self.fp.flat.data_ptr()
        if sp.thin_i or sp.thin_o:
            sp.thin_ws = torch.empty(ops.thin_wgrad_workspace_bytes(), dtype=torch.uint8, device=device)
        gA, gB, gC, gD = new(H, W, 64), new(H, W, 64), new(H, W, 128), new(H, W, 64)
        h1, h2, h3, h4 = new(H // 2, W // 2, 128), new(H // 2, W // 2, 128), new(H // 2, W // 2, 256), new(H // 2, W // 2, 128)
        q1, q2, q3, q4 = new(H // 4, W // 4, 256), new(H // 4, W // 4, 256), new(H // 4, W // 4, 512), new(H // 4, W // 4, 256)
        e1 = new(H // 8, W // 8, 512)
        sp.dxp = new(H, W, 4, dtype=torch.float32)
        sp.keep = [gA, gB, gC, gD, h1, h2, h3, h4, q1, q2, q3, q4, e1]
        ws_bytes = 0

        def wg(name, x, dy, h, w, cin, cout, sn_index=None, k=3, s=1, cin_real=None, cout_real=None, bias=False):
            """weight gradient: plain params write straight into the flat gradient; SN layers write
            dL/d(W/sigma) into the temp buffer and srganfd_spectral_norm_grad finishes the job."""
            nonlocal ws_bytes
            pname = f"{name}.0.weight_orig" if sn_index is not None else f"{name}.weight"
            conv = dict(cin=cin, cout=cout, dw_off=self._poff(pname), db_off=(self._poff(f"{name}.bias") if bias else -1),
                        co_dst=cout_real or cout, ci_dst=cin_real or cin)
            plan = ops.WgradPlan(device, dtc, N, h, w, cin, cout, [conv], ksize=k, stride=s, pad=1)
            ws_bytes = max(ws_bytes, plan.workspace_bytes)
            return ("wgrad", plan, V(x), V(dy), sn_index, name)

        cv = lambda *a, **k: ("conv", ops.conv_args(dtc, *a, **k))
        rs = lambda op, a, b, h, w, c: ("call", lambda: A.check(L.srganfd_resample(op, a, b, dtc, N, h, w, c, A.stream_ptr()), "resample"))
        # bilinear-x2 backward + LeakyReLU' of the upsampled layer in one pass: raw gradient (the skip connection's share) and masked one
        rsl = lambda dy, raw, act, masked, h, w, c: ("call", lambda: A.check(
            L.srganfd_resample_bwd_lrelu(dy, raw, act, masked, dtc, N, h, w, c, 0.2, A.stream_ptr()), "resample_bwd_lrelu"))

        def s2_dgrad(name, dy, dx, hd, wd, cout, cin, r1, mask):
            """data gradient of a 4x4 stride-2 conv: 4 output-parity classes (2x2-tap convs over dy)"""
            items = []
            one = ops.class4_ok(dtc, cin, [O[("b", name, c)] for c in range(4)], ops.packed_bytes(dtc, 2, cout, cin))
            for par in range(1 if one else 4):
                py, px = par >> 1, par & 1
                a = ops.conv_args(dtc, V(dy), V(dx), wptr + O[("b", name, par)], N, hd, wd, cout, cin, ksize=2, stride=1, pad=0,
                                  r1=V(r1) if r1 is not None else A.NULL_VIEW, r1_scale=1.0 if r1 is not None else 0.0,
                                  mask=V(mask) if mask is not None else A.NULL_VIEW, mask_slope=0.2)
                a.h_out, a.w_out = hd, wd
                a.out_sy, a.out_sx, a.out_oy, a.out_ox = 2, 2, py, px
                a.out_h_full, a.out_w_full = 2 * hd, 2 * wd
                a.pad_y, a.pad_x = (1 if py == 0 else 0), (1 if px == 0 else 0)
                a.out_classes, a.class_pad_step = (4, 1) if one else (0, 0)       # one launch: the four classes' workgroups share each dy patch through L2
                items.append(("conv", a))
            return items

        P = N * H * W
        if sp.thin_o:
            w4raw = fptr + 4 * self._poff("conv4.weight")
            head = [("thin", ops.ThinLaunch("thin_wgrad", ops.thin_args(dtc, N, H, W, self.out_ch, w4raw, V(sp.c3), w_big_is_cout=False, thin=sp.dl),
                                            dw_off=self._poff("conv4.weight"), db_off=self._poff("conv4.bias"), ws=sp.thin_ws)),
                    ("thin", ops.ThinLaunch("thin_in", ops.thin_args(dtc, N, H, W, self.out_ch, w4raw, V(gA), w_big_is_cout=False, flip=True, mask=V(sp.c3),
                                                                     mask_slope=0.2, thin=sp.dl)))]
        else:
            head = [wg("conv4", sp.c3, sp.dl, H, W, 64, 32, cout_real=self.out_ch, bias=True),
                    cv(V(sp.dl), V(gA), wptr + O[("b", "conv4")], N, H, W, 32, 64, mask=V(sp.c3), mask_slope=0.2)]
        bw = head + [
            wg("conv3", sp.c2, gA, H, W, 64, 64, sn_index=7),
            cv(V(gA), V(gB), wptr + O[("b", "conv3")], N, H, W, 64, 64, mask=V(sp.c2), mask_slope=0.2),
            wg("conv2", sp.u3, gB, H, W, 64, 64, sn_index=6),
            # u3 = lrelu(z3) + out1: the conv's two outputs are d u3 (y2: the skip's share, added into d out1 below) and
            # d z3 = d u3 * lrelu'(a3) (y, after the mask) -- no separate LeakyReLU-backward pass over the 512^2 tensor
            cv(V(gB), V(gD), wptr + O[("b", "conv2")], N, H, W, 64, 64, y2=V(gA), mask=V(sp.a3), mask_slope=0.2),   # gA = d u3, gD = d z3
            wg("up_block3", sp.b1, gD, H, W, 128, 64, sn_index=5),
            cv(V(gD), V(gC), wptr + O[("b", "up_block3")], N, H, W, 64, 128),            # gC = d b1
            rsl(V(gC), V(h1), V(sp.a2), V(h2), H // 2, W // 2, 128),                     # h1 = d u2, h2 = d z2
            wg("up_block2", sp.b2, h2, H // 2, W // 2, 256, 128, sn_index=4),
            cv(V(h2), V(h3), wptr + O[("b", "up_block2")], N, H // 2, W // 2, 128, 256),  # h3 = d b2
            rsl(V(h3), V(q1), V(sp.a1), V(q2), H // 4, W // 4, 256),                     # q1 = d u1, q2 = d z1
            wg("up_block1", sp.b3, q2, H // 4, W // 4, 512, 256, sn_index=3),
            cv(V(q2), V(q3), wptr + O[("b", "up_block1")], N, H // 4, W // 4, 256, 512),  # q3 = d b3
            rsl(V(q3), A.NULL_VIEW, V(sp.d3), V(e1), H // 8, W // 8, 512),               # e1 = d d3 (pre-activation)
            wg("down_block3", sp.d2, e1, H // 4, W // 4, 256, 512, sn_index=2, k=4, s=2),
        ]
        bw += s2_dgrad("down_block3", e1, q4, H // 8, W // 8, 512, 256, q1, sp.d2)       # q4 = (d d2 + d u1) * lrelu'(d2)
        bw.append(wg("down_block2", sp.d1, q4, H // 2, W // 2, 128, 256, sn_index=1, k=4, s=2))
        bw += s2_dgrad("down_block2", q4, h4, H // 4, W // 4, 256, 128, h1, sp.d1)       # h4 = (d d1 + d u2) * lrelu'(d1)
        bw.append(wg("down_block1", sp.out1, h4, H, W, 64, 128, sn_index=0, k=4, s=2))
        bw += s2_dgrad("down_block1", h4, gD, H // 2, W // 2, 128, 64, gA, None)         # gD = d out1 (+ skip d u3)
        w1raw = fptr + 4 * self._poff("conv1.weight")
        if sp.thin_i:
            bw.append(("thin", ops.ThinLaunch("thin_wgrad", ops.thin_args(dtc, N, H, W, self.in_ch, w1raw, V(gD), w_big_is_cout=True, thin=sp.xin),
                                              dw_off=self._poff("conv1.weight"), db_off=self._poff("conv1.bias"), ws=sp.thin_ws)))
            sp.dx_conv = ops.ThinLaunch("thin_out", ops.thin_args(dtc, N, H, W, self.in_ch, w1raw, V(gD), w_big_is_cout=True, flip=True, thin_out=sp.dxp,
                                                                  thin_out_pitch=4))
        else:
            bw.append(wg("conv1", sp.xin, gD, H, W, 32, 64, cin_real=self.in_ch, bias=True))
            sp.dx_conv = ops.conv_args(dtc, V(gD), V(sp.dxp), wptr + O[("b", "conv1")], N, H, W, 64, 32, cout_store=self.in_ch, y_f32=True)
        sp.bw = bw
        sp.wg_ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
        sp.gtmp = torch.zeros(self.fp.total, dtype=torch.float32, device=device)
        sp.sn_ws = torch.empty(len(SN_LAYERS) * A.SN_GRAD_WS_FLOATS, dtype=torch.float32, device=device)

    # ---- execution ----
    def forward(self, x: Tensor, training: bool) -> Tensor:
        _require_gpu(x)
        dt, dtc = _dt(self.owner)
        dev = x.device
        pk = self._ensure_packed(dtc, dev)
        self._spectral_norm_and_pack(pk, training)
        N, _, H, W = x.shape
        sp = self._plan(N, H, W, dt, dtc, dev, pk)
        L, st = A.lib(), A.stream_ptr()
        x = x.contiguous().float()
        A.check(L.srganfd_nchw_to_nhwc(x.data_ptr(), N, self.in_ch, H, W, A.view(sp.xin), dtc, sp.xin.shape[-1], None, None, st), "nchw_to_nhwc")
        logits = torch.empty(N, self.out_ch, H, W, dtype=torch.float32, device=dev)
        if self.out_ch != 1:
            raise A.SrganfdError("DiscriminatorUNet out_channels must be 1 (logits are written NCHW == NHWC)")
        rec = profiling.REC
        for kind, item in sp.fw + [sp.conv4(logits)]:
            if kind == "thin":
                item.launch(rec)
            elif kind == "conv":
                if rec is None:
                    rc = L.srganfd_conv2d(C.byref(item), st)
                    if rc:
                        A.check(rc, "conv2d")
                else:
                    rec.bracket(profiling.conv_label(item), profiling.conv_work(item), lambda: A.check(L.srganfd_conv2d(C.byref(item), st), "conv2d"))
            else:
                item()
        self.token += 1
        sp.token = self.token
        sp.inv_sigma = pk["scalars"]
        self._last = sp
        return logits

    def backward(self, sp: _Shape, token: int, dlogits: Tensor, need_wgrad: bool, need_dx: bool) -> Tuple[Optional[Tensor], Optional[Tensor]]:
        if getattr(sp, "token", None) != token:
            raise A.SrganfdError("discriminator activations / spectral-norm state were overwritten by a later forward before backward ran")
        L, st = A.lib(), A.stream_ptr()
        N, H, W, dtc = sp.N, sp.H, sp.W, sp.dtc
        dlogits = dlogits.contiguous().float()
        A.check(L.srganfd_nchw_to_nhwc(dlogits.data_ptr(), N, 1, H, W, A.view(sp.dl), dtc, sp.dl.shape[-1], None, None, st), "nchw_to_nhwc")
        flat = self.fp.flat
        flat_grad = self.fp.new_grad(sp.device) if need_wgrad else None
        rec = profiling.REC
        sn_grads = []
        for item in sp.bw:
            kind = item[0]
            if kind == "conv":
                a = item[1]
                if rec is None:
                    rc = L.srganfd_conv2d(C.byref(a), st)
                    if rc:
                        A.check(rc, "conv2d(dgrad)")
                else:
                    rec.bracket(profiling.conv_label(a), profiling.conv_work(a), lambda: A.check(L.srganfd_conv2d(C.byref(a), st), "conv2d(dgrad)"))
            elif kind == "thin":
                if item[1].is_wgrad and not need_wgrad:
                    continue
                item[1].launch(rec, flat_grad.data_ptr() if flat_grad is not None else 0)
            elif kind == "wgrad":
                if not need_wgrad:
                    continue
                _, plan, xv, dyv, sn_index, name = item
                dst = flat_grad if sn_index is None else sp.gtmp
                run = lambda: A.check(L.srganfd_conv2d_wgrad(plan.host, plan.dev.data_ptr(), xv, dyv, dst.data_ptr(), None, sp.wg_ws.data_ptr(),
                                                             sp.wg_ws.numel(), st), "conv2d_wgrad")
                if rec is None:
                    run()
                else:
                    rec.bracket(plan.label, (plan.flops, plan.nbytes), run)
                if sn_index is not None:
                    co, ci, k, _ = self.dims[name]
                    off = 4 * self._poff(f"{name}.0.weight_orig")
                    m = getattr(self.owner, name)[0]
                    sn_grads.append((sp.gtmp.data_ptr() + off, flat.data_ptr() + off, m.weight_u.data_ptr(), m.weight_v.data_ptr(),
                                     sp.inv_sigma.data_ptr() + 4 * (2 * sn_index + 1), flat_grad.data_ptr() + off, co, ci * k * k))
            else:
                item[1]()
        # dL/d(W/sigma) of every spectral-normalised layer sits in its own range of gtmp: one batched pass turns them into dL/dW_orig
        ops.spectral_norm_grad_batch(sn_grads, sp.sn_ws)
        dx = None
        if need_dx:
            if type(sp.dx_conv) is ops.ThinLaunch:
                sp.dx_conv.launch(rec)
            else:
                A.check(L.srganfd_conv2d(C.byref(sp.dx_conv), st), "conv2d(dgrad conv1)")
            dx = torch.empty(N, self.in_ch, H, W, dtype=torch.float32, device=sp.device)
            A.check(L.srganfd_nhwc_to_nchw(A.view(sp.dxp), A.F32, N, self.in_ch, H, W, dx.data_ptr(), 0, st), "nhwc_to_nchw")
        return flat_grad, dx


class _DiscFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eng, training, *params):
        out = eng.forward(x, training)
        ctx.eng, ctx.sp, ctx.token = eng, eng._last, eng.token
        ctx.need_dx = ctx.needs_input_grad[0]
        ctx.need_w = any(ctx.needs_input_grad[3:])
        return out

    @staticmethod
    def backward(ctx, dlogits):
        g, dx = ctx.eng.backward(ctx.sp, ctx.token, dlogits, ctx.need_w, ctx.need_dx)
        grads = tuple(ctx.eng.fp.grad_views(g)) if g is not None else tuple(None for _ in ctx.eng.fp.params)
        return (dx, None, None) + grads


def discriminator_engine(owner: nn.Module) -> DiscriminatorEngine:
    return _engine(owner, lambda: DiscriminatorEngine(owner))


def discriminator_apply(owner: nn.Module, x: Tensor) -> Tensor:
    eng = discriminator_engine(owner)
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in eng.fp.params)):
        return _DiscFn.apply(x, eng, owner.training, *eng.fp.params)
    return eng.forward(x, owner.training)


# ------------------------------------------------------------------------------------------------
# VGG-19 content loss (forward only: the reference detaches the result, model.py:552)
# ------------------------------------------------------------------------------------------------
class ContentLossEngine:
    def __init__(self, owner: nn.Module):
        self.owner = owner
        self.convs = [(int(n), m) for n, m in owner.features.named_children() if isinstance(m, nn.Conv2d)]
        self.pools = [int(n) for n, m in owner.features.named_children() if isinstance(m, nn.MaxPool2d)]
        self.fp = FlatParams([(f"features.{i}.{k}", getattr(m, k)) for i, m in self.convs for k in ("weight", "bias")])
        self.want = [int(n.split(".")[1]) for n in owner.feature_model_extractor_nodes]
        self.shapes = PlanCache()
        self.packed: Dict[int, dict] = {}

    def _ensure_packed(self, dtc, device):
        flat = self.fp.sync(device)
        pk = self.packed.get(dtc)
        if pk is None or pk["buf"].device != device or pk.get("flat_ptr") != flat.data_ptr():
            jobs, offs, cur = [], {}, 0
            for i, m in self.convs:
                co, ci = m.weight.shape[:2]
                offs[i] = cur
                jobs.append(ops.pack_job(cur, dtc, 3, ops.pad32(ci), co, [dict(src_off=self.fp.off(f"features.{i}.weight"), co_src=co, ci_src=ci,
                                                                              k_len=ops.pad32(ci))]))
                cur += (ops.packed_bytes(dtc, 3, ops.pad32(ci), co) + 255) // 256 * 256
            pk = dict(table=ops.PackTable(jobs, device), offs=offs, buf=torch.empty(cur, dtype=torch.uint8, device=device), flat_ptr=flat.data_ptr())
            self.packed[dtc] = pk
        if self.fp.stale(pk):
            pk["table"].run(flat, pk["buf"])
        return pk

    def forward(self, sr: Tensor, gt: Tensor) -> Tensor:
        _require_gpu(sr)
        dt, dtc = _dt(self.owner)
        dev = sr.device
        pk = self._ensure_packed(dtc, dev)
        N, Cin, H, W = sr.shape
        if H < 16 or W < 16:
            raise A.SrganfdError("ContentLoss input height/width must be at least 16 (four 2x2 max-pools; odd sizes floor like torch)")
        L, st = A.lib(), A.stream_ptr()
        key = (N, H, W, dtc, str(dev), pk["buf"].data_ptr(), self.fp.flat.data_ptr())
        sp = self.shapes.get(key)
        if sp is None:
            sp = _Shape()
            # features.0 (3 -> 64) on the thin-side kernel in the 16-bit modes: the normalised image is NHWC with a 4-channel pitch
            sp.thin = ops.thin_ok(dtc, self.owner.features[0].weight.shape[0], Cin)
            sp.xin = torch.empty(2 * N, H, W, 4 if sp.thin else 32, dtype=dt, device=dev)
            sp.bufs = {}
            sp.ws = torch.empty(A.LOSS_WS_FLOATS, dtype=torch.float32, device=dev)
            sp.dt, sp.dtc = dt, dtc
            self.shapes[key] = sp
        self._last = sp
        mean, std = self.owner.mean, self.owner.std
        cpad = sp.xin.shape[-1]
        for img, half in ((sr, 0), (gt, 1)):
            img = img.detach().contiguous().float()
            dst = A.View(sp.xin.data_ptr() + half * N * H * W * cpad * sp.xin.element_size(), cpad, 0)
            A.check(L.srganfd_nchw_to_nhwc(img.data_ptr(), N, Cin, H, W, dst, dtc, cpad, mean.data_ptr(), std.data_ptr(), st), "nchw_to_nhwc")
        losses = torch.zeros(len(self.want), dtype=torch.float32, device=dev)
        last = max(self.want)
        post = self.owner.taps_post_relu
        cur, ch, h, w = sp.xin, 32, H, W
        rec = profiling.REC

        def buf(tag, hh, ww, cc):
            b = sp.bufs.get((tag, hh, ww, cc))
            if b is None:
                b = torch.empty(2 * N, hh, ww, cc, dtype=dt, device=dev)
                sp.bufs[(tag, hh, ww, cc)] = b
            return b
        flip = 0
        for idx in range(last + 1):
            m = self.owner.features[idx]
            if isinstance(m, nn.Conv2d):
                co = m.weight.shape[0]
                out = buf(flip, h, w, co)
                flip ^= 1
                tap = idx in self.want
                # taps are observed after the in-place ReLU unless they are the last requested node
                relu_in_conv = not (tap and (idx == last or not post))
                if idx == 0 and sp.thin:
                    fl = self.fp.flat.data_ptr()
                    ops.ThinLaunch("thin_in", ops.thin_args(dtc, 2 * N, h, w, Cin, fl + 4 * self.fp.off("features.0.weight"), A.view(out), w_big_is_cout=True,
                                                            bias=fl + 4 * self.fp.off("features.0.bias"), act=A.ACT_RELU if relu_in_conv else A.ACT_NONE,
                                                            thin=sp.xin)).launch(rec)
                    a = None
                else:
                    a = ops.conv_args(dtc, A.view(cur), A.view(out), pk["buf"].data_ptr() + pk["offs"][idx], 2 * N, h, w, ch, co,
                                      bias=self.fp.flat.data_ptr() + 4 * self.fp.off(f"features.{idx}.bias"),
                                      act=A.ACT_RELU if relu_in_conv else A.ACT_NONE)
                if a is None:
                    pass
                elif rec is None:
                    A.check(L.srganfd_conv2d(C.byref(a), st), "conv2d(vgg)")
                else:
                    rec.bracket(profiling.conv_label(a), profiling.conv_work(a), lambda: A.check(L.srganfd_conv2d(C.byref(a), st), "conv2d(vgg)"))
                if tap:
                    half_b = N * h * w * co * out.element_size()
                    A.check(L.srganfd_l1_loss_views(A.View(out.data_ptr(), co, 0), A.View(out.data_ptr() + half_b, co, 0), dtc, N * h * w, co, 0, 1.0,
                                                    losses.data_ptr() + 4 * self.want.index(idx), 0, sp.ws.data_ptr(), st), "l1_views")
                    if not relu_in_conv and idx != last:
                        A.check(L.srganfd_resample(4, A.view(out), A.view(out), dtc, 2 * N, h, w, co, st), "relu")
                cur, ch = out, co
            elif isinstance(m, nn.MaxPool2d):
                out = buf("p", h // 2, w // 2, ch)
                A.check(L.srganfd_resample(3, A.view(cur), A.view(out), dtc, 2 * N, h, w, ch, st), "maxpool")
                cur, h, w = out, h // 2, w // 2
        return losses.view(1, -1)


def content_loss_apply(owner: nn.Module, sr: Tensor, gt: Tensor) -> Tensor:
    eng = _engine(owner, lambda: ContentLossEngine(owner))
    with torch.no_grad():
        return eng.forward(sr, gt)
