"""HIP engine for ESRGAN's VGG-style discriminator.

Reference: Discriminator ESRGAN/model.py:88-141 -- conv3x3(bias) + LeakyReLU, nine (conv without bias, BatchNorm2d,
LeakyReLU(0.2)) stages alternating 4x4 stride 2 / 3x3 stride 1 (3x128x128 -> 512x4x4), flatten, Linear(8192,100),
LeakyReLU, Linear(100,1).

The relativistic step of ESRGAN/train_esrgan.py:395-418 keeps up to three training forwards alive at once (gt_output and
sr_output both feed each loss term, ``backward(retain_graph=True)``), each with its own BatchNorm batch statistics, so
activations live in a ring of RING plan instances per input shape instead of one.

Mapping: the convs are the implicit-GEMM kernel (4x4 stride-2 data gradients as four output-parity classes); BatchNorm +
LeakyReLU is one fused statistics/apply pass per stage (srganfd_batchnorm_act_fwd / _bwd, channel blocks of 256);
``torch.flatten`` of an NCHW tensor followed by Linear(8192,100) IS a 4x4 "valid" convolution of the 4x4x512 map with
the weight viewed as (100,512,4,4) -- same bytes -- so the classifier runs on the conv / wgrad kernels too (4x4 stride 2
without padding, one output pixel per image), and Linear(100,1) is a 1x1 conv.  No transposes, no separate GEMM.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor, nn

from . import _abi as A
from . import ops
from . import profiling
from .engine import FlatParams, _dt, _engine, _require_gpu, _Shape, PlanCache

CONV_IDX = (0, 2, 5, 8, 11, 14, 17, 20, 23, 26)          # positions of the convs inside `features`
SLOPE = 0.2
RING = 4                                                   # activation sets per input shape (live forwards of one iteration)


class EsrganDiscriminatorEngine:
    def __init__(self, owner: nn.Module):
        self.owner = owner
        self.fp = FlatParams(list(owner.named_parameters()))
        self.convs = []            # (features index, cin, cout, ksize, stride)
        for fi in CONV_IDX:
            m = owner.features[fi]
            self.convs.append((fi, m.in_channels, m.out_channels, m.kernel_size[0], m.stride[0]))
        if self.convs[0][1] != 3 or owner.classifier[0].in_features != 512 * 16 or owner.classifier[2].out_features != 1:
            raise A.SrganfdError("Discriminator: unexpected layer sizes (ESRGAN/model.py:88-141)")
        self.hid = owner.classifier[0].out_features            # 100
        self.hid_pad = ops.pad32(self.hid)
        self.shapes = PlanCache()
        self.packed: Dict[int, dict] = {}
        self.token = 0

    def _poff(self, n):
        return self.fp.off(n)

    def _build_pack(self, dtc, device):
        jobs, offs, cur = [], {}, 0

        def add(key, ksize, k, n, seg):
            nonlocal cur
            offs[key] = cur
            jobs.append(ops.pack_job(cur, dtc, ksize, k, n, [seg]))
            cur += (ops.packed_bytes(dtc, ksize, k, n) + 255) // 256 * 256

        def layer(key, src, co, ci, ks, stride, fwd=True):
            cip, cop = ops.pad32(ci), ops.pad32(co)
            if fwd:
                add(("f", key), ks, cip, cop, dict(src_off=src, co_src=co, ci_src=ci, k_len=cip))
            if stride == 1:
                add(("b", key), ks, cop, cip, dict(src_off=src, co_src=co, ci_src=ci, k_len=cop, transposed=1))
            else:
                for par in range(4):
                    add(("b", key, par), 2, cop, cip, dict(src_off=src, co_src=co, ci_src=ci, k_len=cop, transposed=2 + par))
        for fi, ci, co, ks, st in self.convs:
            layer(fi, self._poff(f"features.{fi}.weight"), co, ci, ks, st)
        layer("fc1", self._poff("classifier.0.weight"), self.hid, 512, 4, 2)          # Linear(8192,100) == conv 4x4 over the 4x4 map
        layer("fc2", self._poff("classifier.2.weight"), 1, self.hid, 1, 1)
        return dict(table=ops.PackTable(jobs, device), offs=offs, buf=torch.empty(cur, dtype=torch.uint8, device=device))

    def _ensure_packed(self, dtc, device):
        flat = self.fp.sync(device)
        pk = self.packed.get(dtc)
        if pk is None or pk["buf"].device != device or pk.get("flat_ptr") != flat.data_ptr():
            pk = self._build_pack(dtc, device)
            pk["flat_ptr"] = flat.data_ptr()
            self.packed[dtc] = pk
        pk["table"].run(flat, pk["buf"])
        return pk

    # ---- per-shape plan ----
    def _plan(self, N, H, W, dt, dtc, device, pk):
        self._fw_count = getattr(self, "_fw_count", 0) + 1
        key = (N, H, W, dtc, str(device), pk["buf"].data_ptr(), self.fp.flat.data_ptr(), self._fw_count % RING)
        sp = self.shapes.get(key)
        if sp is not None:
            return sp
        if H != 128 or W != 128:
            raise A.SrganfdError("Discriminator expects 3x128x128 inputs: its classifier is Linear(512*4*4, 100) (ESRGAN/model.py:129)")
        sp = _Shape()
        sp.N, sp.H, sp.W, sp.dt, sp.dtc, sp.device = N, H, W, dt, dtc, device
        V = A.view
        fptr, wptr, O = self.fp.flat.data_ptr(), pk["buf"].data_ptr(), pk["offs"]

        def new(h, w, c, dtype=dt, zero=False):
            f = torch.zeros if zero else torch.empty
            return f(N, h, w, c, dtype=dtype, device=device)
        sp.xin = new(H, W, 32)
        sp.y, sp.a, sp.save, sp.hw = {}, {}, {}, {}
        h, w = H, W
        fw = []
        cv = lambda *a, **k: ("conv", ops.conv_args(dtc, *a, **k))
        prev = sp.xin
        for i, (fi, ci, co, ks, st) in enumerate(self.convs):
            ho, wo = h // st, w // st
            sp.hw[i] = (h, w, ho, wo)
            sp.a[i] = new(ho, wo, co)
            if i == 0:
                fw.append(cv(V(prev), V(sp.a[0]), wptr + O[("f", fi)], N, h, w, 32, co, bias=fptr + 4 * self._poff("features.0.bias"),
                             act=A.ACT_LRELU, slope=SLOPE))
            else:
                sp.y[i] = new(ho, wo, co)
                sp.save[i] = torch.empty(4 * co, dtype=torch.float32, device=device)
                fw.append(cv(V(prev), V(sp.y[i]), wptr + O[("f", fi)], N, h, w, ci, co, ksize=ks, stride=st))
                fw.append(("bn", i))
            prev, h, w = sp.a[i], ho, wo
        sp.f1 = new(1, 1, self.hid_pad, zero=True)          # padded channels stay zero (they meet zero weights in fc2)
        a = ops.conv_args(dtc, V(sp.a[9]), V(sp.f1), wptr + O[("f", "fc1")], N, 4, 4, 512, self.hid_pad, cout_store=self.hid, ksize=4, stride=2,
                          pad=0, bias=fptr + 4 * self._poff("classifier.0.bias"), act=A.ACT_LRELU, slope=SLOPE)
        fw.append(("conv", a))
        sp.fw = fw
        f1_v, hid, w2, b2 = V(sp.f1), self.hid_pad, wptr + O[("f", "fc2")], fptr + 4 * self._poff("classifier.2.bias")   # no `sp` / `self` in the closure
        sp.fc2 = lambda logits: ops.conv_args(dtc, f1_v, A.View(logits.data_ptr(), 1, 0), w2, N, 1, 1, hid, 32,
                                               cout_store=1, ksize=1, pad=0, bias=b2, y_f32=True)
        sp.bn_ws = torch.empty(2048 * 256 + 3 * 256, dtype=torch.float32, device=device)
        self._plan_backward(sp, pk)
        self.shapes[key] = sp
        return sp

    def _plan_backward(self, sp, pk):
        N, dt, dtc, device = sp.N, sp.dt, sp.dtc, sp.device
        V = A.view
        wptr, O = pk["buf"].data_ptr(), pk["offs"]

        def new(h, w, c, dtype=dt):
            return torch.empty(N, h, w, c, dtype=dtype, device=device)
        ws_bytes = 0

        def wg(pname, bname, x, dy, h, w, cin, cout, k, s, pad, cin_real=None, cout_real=None):
            nonlocal ws_bytes
            conv = dict(cin=cin, cout=cout, dw_off=self._poff(pname), db_off=(self._poff(bname) if bname else -1),
                        co_dst=cout_real or cout, ci_dst=cin_real or cin)
            plan = ops.WgradPlan(device, dtc, N, h, w, cin, cout, [conv], ksize=k, stride=s, pad=pad)
            ws_bytes = max(ws_bytes, plan.workspace_bytes)
            return ("wgrad", plan, V(x), V(dy))

        def s2_dgrad(key, dy, dx, hd, wd, cout, cin, mask, pad):
            """data gradient of a 4x4 stride-2 conv as 4 output-parity classes (2x2-tap convs over dy).  pad = 1: class
            (py,px) has hd x wd outputs; pad = 0 (the classifier's 4x4 'valid' conv): hd+1 x wd+1 outputs, the tap pairs of
            the opposite parity and one row/column of zero padding on the low side."""
            items = []
            one = pad == 1 and ops.class4_ok(dtc, cin, [O[("b", key, c)] for c in range(4)], ops.packed_bytes(dtc, 2, cout, cin))
            for par in range(1 if one else 4):
                py, px = par >> 1, par & 1
                wpar = par if pad == 1 else (((1 - py) << 1) | (1 - px))
                a = ops.conv_args(dtc, V(dy), V(dx), wptr + O[("b", key, wpar)], N, hd, wd, cout, cin, ksize=2, stride=1, pad=0,
                                  mask=V(mask) if mask is not None else A.NULL_VIEW, mask_slope=SLOPE)
                ext = 0 if pad == 1 else 1
                a.h_out, a.w_out = hd + ext, wd + ext
                a.out_sy, a.out_sx, a.out_oy, a.out_ox = 2, 2, py, px
                a.out_h_full, a.out_w_full = 2 * hd + 2 * ext, 2 * wd + 2 * ext
                a.pad_y, a.pad_x = ((1 if py == 0 else 0), (1 if px == 0 else 0)) if pad == 1 else (1, 1)
                a.out_classes, a.class_pad_step = (4, 1) if one else (0, 0)
                items.append(("conv", a))
            return items

        cv = lambda *a, **k: ("conv", ops.conv_args(dtc, *a, **k))
        sp.dl = new(1, 1, 32)
        df1 = new(1, 1, self.hid_pad)
        bw = [
            wg("classifier.2.weight", "classifier.2.bias", sp.f1, sp.dl, 1, 1, self.hid_pad, 32, 1, 1, 0, cin_real=self.hid, cout_real=1),
            cv(V(sp.dl), V(df1), wptr + O[("b", "fc2")], N, 1, 1, 32, self.hid_pad, ksize=1, pad=0, mask=V(sp.f1), mask_slope=SLOPE),
            wg("classifier.0.weight", "classifier.0.bias", sp.a[9], df1, 4, 4, 512, self.hid_pad, 4, 2, 0, cout_real=self.hid),
        ]
        dA = new(4, 4, 512)                       # gradient w.r.t. a9 (post-activation): the BN backward applies LeakyReLU'
        bw += s2_dgrad("fc1", df1, dA, 1, 1, self.hid_pad, 512, None, 0)
        sp.keep = [df1, dA]
        for i in range(9, 0, -1):
            fi, ci, co, ks, st = self.convs[i]
            h, w, ho, wo = sp.hw[i]
            dY = new(ho, wo, co)
            bw.append(("bn_bwd", i, V(dA), V(dY)))
            xprev = sp.a[i - 1]
            bw.append(wg(f"features.{fi}.weight", None, xprev, dY, h, w, ci, co, ks, st, 1))
            dAp = new(h, w, ci)
            # below stage 1 sits conv0 + LeakyReLU (no BatchNorm): its activation derivative goes into this epilogue
            mask0 = sp.a[0] if i == 1 else None
            if st == 1:
                bw.append(cv(V(dY), V(dAp), wptr + O[("b", fi)], N, ho, wo, co, ci, mask=V(mask0) if mask0 is not None else A.NULL_VIEW,
                             mask_slope=SLOPE))
            else:
                bw += s2_dgrad(fi, dY, dAp, ho, wo, co, ci, mask0, 1)
            sp.keep += [dY, dAp]
            dA = dAp
        # dA is now dL/d(conv0 output before LeakyReLU)
        bw.append(wg("features.0.weight", "features.0.bias", sp.xin, dA, sp.H, sp.W, 32, 64, 3, 1, 1, cin_real=3))
        sp.bw = bw
        sp.dxp = torch.empty(N, sp.H, sp.W, 4, dtype=torch.float32, device=device)
        sp.dx_conv = ops.conv_args(dtc, V(dA), V(sp.dxp), wptr + O[("b", 0)], N, sp.H, sp.W, 64, 32, cout_store=3, y_f32=True)
        sp.wg_ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
        sp.gtmp = torch.zeros(self.fp.total, dtype=torch.float32, device=device)

    # ---- execution ----
    def _conv(self, L, st, a, rec, what):
        if rec is None:
            rc = L.srganfd_conv2d(C.byref(a), st)
            if rc:
                A.check(rc, what)
        else:
            rec.bracket(profiling.conv_label(a), profiling.conv_work(a), lambda: A.check(L.srganfd_conv2d(C.byref(a), st), what))

    def forward(self, x: Tensor, training: bool) -> Tensor:
        _require_gpu(x)
        dt, dtc = _dt(self.owner)
        dev = x.device
        pk = self._ensure_packed(dtc, dev)
        N, _, H, W = x.shape
        sp = self._plan(N, H, W, dt, dtc, dev, pk)
        L, st = A.lib(), A.stream_ptr()
        x = x.contiguous().float()
        A.check(L.srganfd_nchw_to_nhwc(x.data_ptr(), N, 3, H, W, A.view(sp.xin), dtc, 32, None, None, st), "nchw_to_nhwc")
        logits = torch.empty(N, 1, dtype=torch.float32, device=dev)
        rec = profiling.REC
        flat = self.fp.flat
        for kind, item in sp.fw + [("conv", sp.fc2(logits))]:
            if kind == "conv":
                self._conv(L, st, item, rec, "conv2d")
            else:                                           # BatchNorm2d + LeakyReLU of stage `item`
                i = item
                fi, ci, co, ks, s_ = self.convs[i]
                bn = self.owner.features[fi + 1]
                if bn.running_mean.device != dev:
                    raise A.SrganfdError("BatchNorm buffers must live on the module's GPU")
                _, _, ho, wo = sp.hw[i]
                A.check(L.srganfd_batchnorm_act_fwd(A.view(sp.y[i]), A.view(sp.a[i]), dtc, N * ho * wo, co,
                                                    flat.data_ptr() + 4 * self._poff(f"features.{fi + 1}.weight"),
                                                    flat.data_ptr() + 4 * self._poff(f"features.{fi + 1}.bias"),
                                                    bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.momentum, bn.eps,
                                                    1 if training else 0, sp.save[i].data_ptr(), sp.bn_ws.data_ptr(), SLOPE, st), "batchnorm_act_fwd")
                if training:
                    bn.num_batches_tracked += 1
        self.token += 1
        sp.token, sp.training = self.token, training
        self._last = sp
        return logits

    def backward(self, sp, token, dlogits: Tensor, need_wgrad: bool, need_dx: bool) -> Tuple[Optional[Tensor], Optional[Tensor]]:
        if getattr(sp, "token", None) != token:
            raise A.SrganfdError("discriminator activations / BatchNorm statistics were overwritten by a later forward before backward ran")
        if not sp.training:
            raise A.SrganfdError("Discriminator backward is implemented for training-mode forwards (BatchNorm batch statistics)")
        L, st = A.lib(), A.stream_ptr()
        N, dtc = sp.N, sp.dtc
        dlogits = dlogits.contiguous().float()
        A.check(L.srganfd_nchw_to_nhwc(dlogits.data_ptr(), N, 1, 1, 1, A.view(sp.dl), dtc, 32, None, None, st), "nchw_to_nhwc")
        flat = self.fp.flat
        flat_grad = self.fp.new_grad(sp.device) if need_wgrad else sp.gtmp
        rec = profiling.REC
        for item in sp.bw:
            kind = item[0]
            if kind == "conv":
                self._conv(L, st, item[1], rec, "conv2d(dgrad)")
            elif kind == "wgrad":
                if not need_wgrad:
                    continue
                _, plan, xv, dyv = item
                run = lambda: A.check(L.srganfd_conv2d_wgrad(plan.host, plan.dev.data_ptr(), xv, dyv, flat_grad.data_ptr(), None, sp.wg_ws.data_ptr(),
                                                             sp.wg_ws.numel(), st), "conv2d_wgrad")
                if rec is None:
                    run()
                else:
                    rec.bracket(plan.label, (plan.flops, plan.nbytes), run)
            else:                                           # bn_bwd: dA (w.r.t. post-activation) -> dY (w.r.t. conv output)
                _, i, dAv, dYv = item
                fi, ci, co, ks, s_ = self.convs[i]
                _, _, ho, wo = sp.hw[i]
                A.check(L.srganfd_batchnorm_act_bwd(A.view(sp.y[i]), dAv, dYv, dtc, N * ho * wo, co,
                                                    flat.data_ptr() + 4 * self._poff(f"features.{fi + 1}.weight"), sp.save[i].data_ptr(),
                                                    flat_grad.data_ptr() + 4 * self._poff(f"features.{fi + 1}.weight"),
                                                    flat_grad.data_ptr() + 4 * self._poff(f"features.{fi + 1}.bias"), 0.0, sp.bn_ws.data_ptr(),
                                                    A.view(sp.a[i]), SLOPE, st), "batchnorm_act_bwd")
        dx = None
        if need_dx:
            A.check(L.srganfd_conv2d(C.byref(sp.dx_conv), st), "conv2d(dgrad conv0)")
            dx = torch.empty(N, 3, sp.H, sp.W, dtype=torch.float32, device=sp.device)
            A.check(L.srganfd_nhwc_to_nchw(A.view(sp.dxp), A.F32, N, 3, sp.H, sp.W, dx.data_ptr(), 0, st), "nhwc_to_nchw")
        return (flat_grad if need_wgrad else None), dx


class _EsrganDFn(torch.autograd.Function):
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


def esrgan_discriminator_engine(owner: nn.Module) -> EsrganDiscriminatorEngine:
    return _engine(owner, lambda: EsrganDiscriminatorEngine(owner))


def esrgan_discriminator_apply(owner: nn.Module, x: Tensor) -> Tensor:
    eng = esrgan_discriminator_engine(owner)
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in eng.fp.params)):
        return _EsrganDFn.apply(x, eng, owner.training, *eng.fp.params)
    return eng.forward(x, owner.training)
