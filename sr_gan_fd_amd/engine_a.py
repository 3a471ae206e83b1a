"""HIP engine for the A-ESRGAN attention U-Net discriminator (BASELINE config 5).

Reference: UNetDiscriminatorAesrgan.forward A-ESRGAN/model.py:311-338, add_attn.forward :239-254,
unetCat.forward :265-275, spectral_norm as in engine_d.py.

Data flow (NHWC; R0 = input size, Rk = R0 / 2^k, Rg = R3 + 2):
  x -> conv0 -> x0 -3x3s2-> x1 -3x3s2-> x2 -3x3s2-> x3 -1x1 pad1-> gated (Rg)
  attention gate k on (x2 | x1 | x0): theta = conv2x2s2(x); phi = resize(conv1x1(gated)); f = relu(theta+phi);
      sig = sigmoid(conv1x1(f)) (fp32 map); y = up2(sig) * x; BN(conv1x1(y)) -> channels [0,C) of cat_k
  cat_k channels [C,2C) = lrelu(convU_k(up2(prev)));  x4 = conv4(cat_1) ... x6 = conv6(cat_3) -> conv7 -> conv8 -> conv9
torch.cat never runs (both halves are written into one buffer); the skip/attention gradient sums are folded
into the data-gradient epilogues (two residual inputs + LeakyReLU' mask); the stride-2 3x3 data gradients run
as 4 output-parity classes of 2x2-tap convs, the 2x2 stride-2 ones as 4 classes of 1x1 convs.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import weakref

import torch
from torch import Tensor, nn

from . import _abi as A
from . import ops
from . import profiling
from .engine import FlatParams, _dt, _engine, _require_gpu, _Shape, PlanCache

# (name, ksize, stride, pad) of the spectral-normalised convs, in forward order
SN_LAYERS = [("conv1", 3, 2, 1), ("conv2", 3, 2, 1), ("conv3", 3, 2, 1), ("gating", 1, 1, 1), ("cat_1.convU", 3, 1, 1),
             ("conv4", 3, 1, 1), ("cat_2.convU", 3, 1, 1), ("conv5", 3, 1, 1), ("cat_3.convU", 3, 1, 1), ("conv6", 3, 1, 1),
             ("conv7", 3, 1, 1), ("conv8", 3, 1, 1)]
SN_INDEX = {n: i for i, (n, _, _, _) in enumerate(SN_LAYERS)}


def _mod(owner: nn.Module, name: str) -> nn.Module:
    m = owner
    for part in name.split("."):
        m = getattr(m, part)
    return m


class AesrganDiscriminatorEngine:
    def __init__(self, owner: nn.Module):
        self.owner = owner
        self.fp = FlatParams(list(owner.named_parameters()))
        self.nf = owner.conv0.weight.shape[0]
        self.in_ch = owner.conv0.weight.shape[1]
        if self.nf != 64:
            raise A.SrganfdError("UNetDiscriminatorAesrgan: num_feat must be 64 (channel counts are multiples of 32, BatchNorm <= 256 channels)")
        self.shapes = PlanCache()
        self.packed: Dict[int, dict] = {}
        self.token = 0
        self.sync_bn = None   # data parallel: SyncBatchNormReduce -- batch statistics over all ranks (set by GanTrainer(sync_batchnorm=True))

    def _poff(self, n):
        return self.fp.off(n)

    def _wshape(self, name):
        w = _mod(self.owner, name)
        w = w.weight_orig if hasattr(w, "weight_orig") else w.weight
        return w.shape[0], w.shape[1]

    # ---- packing ----
    def _build_pack(self, dtc, device):
        jobs, offs, cur = [], {}, 0

        def add(key, ksize, k, n, seg):
            nonlocal cur
            offs[key] = cur
            jobs.append(ops.pack_job(cur, dtc, ksize, k, n, [seg]))
            cur += (ops.packed_bytes(dtc, ksize, k, n) + 255) // 256 * 256

        def plain(name, ks, fwd=True, bwd=True):
            co, ci = self._wshape(name)
            src = self._poff(name + ".weight")
            if fwd:
                add(("f", name), ks, ops.pad32(ci), ops.pad32(co), dict(src_off=src, co_src=co, ci_src=ci, k_len=ops.pad32(ci)))
            if bwd:
                add(("b", name), ks, ops.pad32(co), ops.pad32(ci), dict(src_off=src, co_src=co, ci_src=ci, k_len=ops.pad32(co), transposed=1))
        plain("conv0", 3)
        plain("conv9", 3)
        for l, (name, ks, st, _) in enumerate(SN_LAYERS):
            co, ci = self._wshape(name)
            src = self._poff(name + ".weight_orig")
            sc = 2 * l + 1
            add(("f", name), ks, ci, co, dict(src_off=src, co_src=co, ci_src=ci, k_len=ci, scale_off=sc))
            if st == 1:
                add(("b", name), ks, co, ci, dict(src_off=src, co_src=co, ci_src=ci, k_len=co, transposed=1, scale_off=sc))
            else:
                for par in range(4):   # 3x3 stride-2 data gradient: 4 parity classes of 2x2-tap convs
                    add(("b", name, par), 2, co, ci, dict(src_off=src, co_src=co, ci_src=ci, k_len=co, transposed=6 + par, scale_off=sc))
        for k in (1, 2, 3):
            pre = f"attn_{k}"
            plain(pre + ".W.0", 1)
            plain(pre + ".phi", 1)
            plain(pre + ".psi", 1)
            co, ci = self._wshape(pre + ".theta")
            src = self._poff(pre + ".theta.weight")
            add(("f", pre + ".theta"), 2, ci, co, dict(src_off=src, co_src=co, ci_src=ci, k_len=ci))
            for ab in range(4):        # 2x2 stride-2 data gradient: 4 classes of 1x1 convs
                add(("b", pre + ".theta", ab), 1, co, ci, dict(src_off=src, co_src=co, ci_src=ci, k_len=co, transposed=10 + ab))
        return dict(table=ops.PackTable(jobs, device), offs=offs, buf=torch.empty(cur, dtype=torch.uint8, device=device))

    def _ensure_packed(self, dtc, device):
        flat = self.fp.sync(device)
        pk = self.packed.get(dtc)
        if pk is None or pk["buf"].device != device or pk.get("flat_ptr") != flat.data_ptr():
            pk = self._build_pack(dtc, device)
            pk["flat_ptr"] = flat.data_ptr()
            pk["scalars"] = torch.ones(2 * len(SN_LAYERS), dtype=torch.float32, device=device)
            pk["sn_ws"] = torch.empty(sum(A.sn_ws_floats(self._wshape(n)[0], self._wshape(n)[1] * ks * ks) for n, ks, _, _ in SN_LAYERS),
                                      dtype=torch.float32, device=device)
            self.packed[dtc] = pk
        return pk

    def _spectral_norm_and_pack(self, pk, training):
        flat = self.fp.flat
        sc = pk["scalars"].data_ptr()
        layers = []
        for l, (name, ks, _, _) in enumerate(SN_LAYERS):
            co, ci = self._wshape(name)
            m = _mod(self.owner, name)
            layers.append((flat.data_ptr() + 4 * self._poff(name + ".weight_orig"), m.weight_u.data_ptr(), m.weight_v.data_ptr(), co, ci * ks * ks,
                           sc + 8 * l, sc + 8 * l + 4))
        ops.spectral_norm_batch(layers, training, pk["sn_ws"])
        pk["table"].run(flat, pk["buf"], pk["scalars"])

    # ---- plan ----
    def _plan(self, N, H, W, dt, dtc, device, pk):
        key = (N, H, W, dtc, str(device), pk["buf"].data_ptr(), self.fp.flat.data_ptr())
        sp = self.shapes.get(key)
        if sp is not None:
            return sp
        if H % 8 or W % 8:
            raise A.SrganfdError("UNetDiscriminatorAesrgan input height/width must be multiples of 8")
        sp = _Shape()
        sp.N, sp.H, sp.W, sp.dt, sp.dtc, sp.device = N, H, W, dt, dtc, device
        V = A.view
        L = A.lib()
        fptr, wptr, O = self.fp.flat.data_ptr(), pk["buf"].data_ptr(), pk["offs"]
        me = weakref.proxy(self)                    # closures stored on the plan must not hold the engine (reference cycle)
        P = lambda name: fptr + 4 * me._poff(name)
        Wp = lambda *key: wptr + O[key]
        nf = self.nf

        def new(h, w, c, dtype=dt):
            return torch.empty(N, h, w, c, dtype=dtype, device=device)
        R = [(H >> k, W >> k) for k in range(4)]
        Hg, Wg = R[3][0] + 2, R[3][1] + 2
        B = sp.B = {}
        # conv0 (in_ch -> nf) and conv9 (nf -> 1) on the thin-side kernels in the 16-bit modes (csrc/conv_thin.hip): 4-channel pitch
        sp.thin_i, sp.thin_o = ops.thin_ok(dtc, nf, self.in_ch), ops.thin_ok(dtc, nf, 1)
        B["xin"] = new(H, W, 4 if sp.thin_i else 32)
        B["x0"], B["x1"], B["x2"], B["x3"] = new(*R[0], nf), new(*R[1], 2 * nf), new(*R[2], 4 * nf), new(*R[3], 8 * nf)
        B["gated"] = new(Hg, Wg, 4 * nf)
        B["cat1"], B["cat2"], B["cat3"] = new(*R[2], 8 * nf), new(*R[1], 4 * nf), new(*R[0], 2 * nf)
        B["b3"], B["x4"], B["b4"], B["x5"], B["b5"] = new(*R[2], 8 * nf), new(*R[2], 4 * nf), new(*R[1], 4 * nf), new(*R[1], 2 * nf), new(*R[0], 2 * nf)
        B["x6"], B["c7"], B["c8"] = new(*R[0], nf), new(*R[0], nf), new(*R[0], nf)
        sp.bn_ws = torch.empty(2048 * 256 + 3 * 256, dtype=torch.float32, device=device)
        sp.bn_ws_global = None
        lre = dict(act=A.ACT_LRELU, slope=0.2)
        cv = lambda *a, **k: ("conv", ops.conv_args(dtc, *a, **k))
        call = lambda fn: ("call", fn)
        st = A.stream_ptr
        rs = lambda op, a, b, h, w, c, dtype=dtc: call(lambda: A.check(L.srganfd_resample(op, a, b, dtype, N, h, w, c, st()), "resample"))
        fw: List[tuple] = [
            ("thin", ops.ThinLaunch("thin_in", ops.thin_args(dtc, N, H, W, self.in_ch, P("conv0.weight"), V(B["x0"]), w_big_is_cout=True, bias=P("conv0.bias"),
                                                             thin=B["xin"], **lre)))
            if sp.thin_i else cv(V(B["xin"]), V(B["x0"]), Wp("f", "conv0"), N, H, W, 32, nf, bias=P("conv0.bias"), **lre),
            cv(V(B["x0"]), V(B["x1"]), Wp("f", "conv1"), N, *R[0], nf, 2 * nf, stride=2, **lre),
            cv(V(B["x1"]), V(B["x2"]), Wp("f", "conv2"), N, *R[1], 2 * nf, 4 * nf, stride=2, **lre),
            cv(V(B["x2"]), V(B["x3"]), Wp("f", "conv3"), N, *R[2], 4 * nf, 8 * nf, stride=2, **lre),
            cv(V(B["x3"]), V(B["gated"]), Wp("f", "gating"), N, *R[3], 8 * nf, 4 * nf, ksize=1, pad=1, **lre),
        ]
        sp.attn = {}
        for k, xname, cat in ((1, "x2", "cat1"), (2, "x1", "cat2"), (3, "x0", "cat3")):
            pre = f"attn_{k}"
            Ck = B[xname].shape[-1]
            h, w = R[3 - k]
            hh, wh = h // 2, w // 2
            T = {}
            T["theta"], T["phi"], T["phiup"], T["f"] = new(hh, wh, Ck), new(Hg, Wg, Ck), new(hh, wh, Ck), new(hh, wh, Ck)
            T["sig"] = torch.empty(N, hh, wh, 1, dtype=torch.float32, device=device)
            T["sigup"] = torch.empty(N, h, w, 1, dtype=torch.float32, device=device)
            T["y"], T["wy"] = new(h, w, Ck), new(h, w, Ck)
            T["save"] = torch.empty(4 * Ck, dtype=torch.float32, device=device)
            T["dims"] = (h, w, hh, wh, Ck)
            sp.attn[k] = T
            bn = _mod(self.owner, pre + ".W.1")
            fw += [
                cv(V(B[xname]), V(T["theta"]), Wp("f", pre + ".theta"), N, h, w, Ck, Ck, ksize=2, stride=2, pad=0),
                cv(V(B["gated"]), V(T["phi"]), Wp("f", pre + ".phi"), N, Hg, Wg, 4 * nf, Ck, ksize=1, pad=0, bias=P(pre + ".phi.bias")),
                call(lambda a=V(T["phi"]), b=V(T["phiup"]), hh=hh, wh=wh, Ck=Ck: A.check(
                    L.srganfd_resize_bilinear(0, a, b, dtc, N, Hg, Wg, hh, wh, Ck, st()), "resize")),
                call(lambda a=V(T["theta"]), b=V(T["phiup"]), o=V(T["f"]), n=N * hh * wh, Ck=Ck: A.check(L.srganfd_add_relu(a, b, o, dtc, n, Ck, st()), "add_relu")),
                cv(V(T["f"]), V(T["sig"]), Wp("f", pre + ".psi"), N, hh, wh, Ck, 32, ksize=1, pad=0, cout_store=1, bias=P(pre + ".psi.bias"), y_f32=True),
                call(lambda t=T["sig"]: A.check(L.srganfd_sigmoid(t.data_ptr(), t.numel(), st()), "sigmoid")),
                rs(1, V(T["sig"]), V(T["sigup"]), hh, wh, 1, A.F32),
                call(lambda x=V(B[xname]), g=T["sigup"], y=V(T["y"]), n=N * h * w, Ck=Ck: A.check(
                    L.srganfd_gate_mul(0, x, g.data_ptr(), y, A.NULL_VIEW, None, dtc, n, Ck, st()), "gate_mul")),
                cv(V(T["y"]), V(T["wy"]), Wp("f", pre + ".W.0"), N, h, w, Ck, Ck, ksize=1, pad=0, bias=P(pre + ".W.0.bias")),
                ("bn", (k, V(T["wy"]), V(B[cat]), N * h * w, Ck, P(pre + ".W.1.weight"), P(pre + ".W.1.bias"), bn, T["save"])),
            ]
        fw += [
            rs(1, V(B["x3"]), V(B["b3"]), *R[3], 8 * nf),
            cv(V(B["b3"]), V(B["cat1"], c0=4 * nf), Wp("f", "cat_1.convU"), N, *R[2], 8 * nf, 4 * nf, **lre),
            cv(V(B["cat1"]), V(B["x4"]), Wp("f", "conv4"), N, *R[2], 8 * nf, 4 * nf, **lre),
            rs(1, V(B["x4"]), V(B["b4"]), *R[2], 4 * nf),
            cv(V(B["b4"]), V(B["cat2"], c0=2 * nf), Wp("f", "cat_2.convU"), N, *R[1], 4 * nf, 2 * nf, **lre),
            cv(V(B["cat2"]), V(B["x5"]), Wp("f", "conv5"), N, *R[1], 4 * nf, 2 * nf, **lre),
            rs(1, V(B["x5"]), V(B["b5"]), *R[1], 2 * nf),
            cv(V(B["b5"]), V(B["cat3"], c0=nf), Wp("f", "cat_3.convU"), N, *R[0], 2 * nf, nf, **lre),
            cv(V(B["cat3"]), V(B["x6"]), Wp("f", "conv6"), N, *R[0], 2 * nf, nf, **lre),
            cv(V(B["x6"]), V(B["c7"]), Wp("f", "conv7"), N, *R[0], nf, nf, **lre),
            cv(V(B["c7"]), V(B["c8"]), Wp("f", "conv8"), N, *R[0], nf, nf, **lre),
        ]
        sp.fw = fw
        if sp.thin_o:
            sp.conv9 = lambda logits: ("thin", ops.ThinLaunch("thin_out", ops.thin_args(dtc, N, H, W, 1, P("conv9.weight"), V(B["c8"]), w_big_is_cout=False,
                                                                                        bias=P("conv9.bias"), thin_out=logits.data_ptr(), thin_out_pitch=1)))
        else:
            sp.conv9 = lambda logits: ("conv", ops.conv_args(dtc, V(B["c8"]), A.View(logits.data_ptr(), 1, 0), Wp("f", "conv9"), N, H, W, nf, 32,
                                                             cout_store=1, bias=P("conv9.bias"), y_f32=True))
        sp.R, sp.Rg = R, (Hg, Wg)
        self._plan_backward(sp, pk)
        self.shapes[key] = sp
        return sp

    def _plan_backward(self, sp, pk):
        N, H, W, dt, dtc, device = sp.N, sp.H, sp.W, sp.dt, sp.dtc, sp.device
        V, L, B, R, nf = A.view, A.lib(), sp.B, sp.R, self.nf
        Hg, Wg = sp.Rg
        wptr, O = pk["buf"].data_ptr(), pk["offs"]
        Wp = lambda *key: wptr + O[key]
        fptr = self.fp.flat.data_ptr()
        P = lambda name: fptr + 4 * self.fp.off(name)      # (used while the plan is built only, not stored on it)
        st = A.stream_ptr

        def new(h, w, c, dtype=dt):
            return torch.empty(N, h, w, c, dtype=dtype, device=device)
        ws_bytes = 0

        def wg(name, x, dy, h, w, cin, cout, k=3, s=1, pad=1, sn=False, cin_real=None, cout_real=None, bias=False, x_c0=0, dy_c0=0):
            nonlocal ws_bytes
            pname = name + (".weight_orig" if sn else ".weight")
            conv = dict(cin=cin, cout=cout, dw_off=self._poff(pname), db_off=(self._poff(name + ".bias") if bias else -1),
                        co_dst=cout_real or cout, ci_dst=cin_real or cin)
            plan = ops.WgradPlan(device, dtc, N, h, w, cin, cout, [conv], ksize=k, stride=s, pad=pad)
            ws_bytes = max(ws_bytes, plan.workspace_bytes)
            return ("wgrad", plan, V(x, c0=x_c0), V(dy, c0=dy_c0), SN_INDEX[name] if sn else None, name, k)

        cv = lambda *a, **k: ("conv", ops.conv_args(dtc, *a, **k))
        call = lambda fn: ("call", fn)
        rs = lambda op, a, b, h, w, c, dtype=dtc: call(lambda: A.check(L.srganfd_resample(op, a, b, dtype, N, h, w, c, st()), "resample"))
        lb = lambda dy, act, out, npix, c, slope=0.2: call(lambda: A.check(L.srganfd_lrelu_bwd(dy, act, A.NULL_VIEW, out, dtc, npix, c, slope, st()), "lrelu_bwd"))

        def strided_dgrad(key_fn, ks, dy, dx, hd, wd, cin_op, cout_op, r1=None, r2=None, mask=None):
            """4 parity classes writing a (2hd x 2wd) image: ks=2 (3x3 s2 conv) or ks=1 (2x2 s2 conv)"""
            items = []
            one = ops.class4_ok(dtc, cout_op, [O[key_fn(c)] for c in range(4)], ops.packed_bytes(dtc, ks, cin_op, cout_op), ksize=ks)
            for par in range(1 if one else 4):
                py, px = par >> 1, par & 1
                a = ops.conv_args(dtc, V(dy), V(dx), Wp(*key_fn(par)), N, hd, wd, cin_op, cout_op, ksize=ks, stride=1, pad=0,
                                  r1=V(r1) if r1 is not None else A.NULL_VIEW, r1_scale=1.0 if r1 is not None else 0.0,
                                  r2=V(r2) if r2 is not None else A.NULL_VIEW, r2_scale=1.0 if r2 is not None else 0.0,
                                  mask=V(mask) if mask is not None else A.NULL_VIEW, mask_slope=0.2)
                a.h_out, a.w_out = hd, wd
                a.out_sy, a.out_sx, a.out_oy, a.out_ox = 2, 2, py, px
                a.out_h_full, a.out_w_full = 2 * hd, 2 * wd
                a.pad_y, a.pad_x = 0, 0
                a.out_classes, a.class_pad_step = (4, 0) if one else (0, 0)     # one launch: every class reads the same window of dy
                items.append(("conv", a))
            return items

        sp.dl = new(H, W, 4 if sp.thin_o else 32)
        if sp.thin_i or sp.thin_o:
            sp.thin_ws = torch.empty(ops.thin_wgrad_workspace_bytes(), dtype=torch.uint8, device=device)
        G = sp.G = {}
        G["g8"], G["g7"], G["g6"] = new(*R[0], nf), new(*R[0], nf), new(*R[0], nf)
        G["dc3"], G["db5"], G["dx5"] = new(*R[0], 2 * nf), new(*R[0], 2 * nf), new(*R[1], 2 * nf)
        G["dc2"], G["db4"], G["dx4"] = new(*R[1], 4 * nf), new(*R[1], 4 * nf), new(*R[2], 4 * nf)
        G["dc1"], G["db3"], G["dx3a"], G["dx3"] = new(*R[2], 8 * nf), new(*R[2], 8 * nf), new(*R[3], 8 * nf), new(*R[3], 8 * nf)
        G["dx2"], G["dx1"], G["dx0"] = new(*R[2], 4 * nf), new(*R[1], 2 * nf), new(*R[0], nf)
        G["dgated"] = [new(Hg, Wg, 4 * nf) for _ in range(3)]
        sp.dxp = new(H, W, 4, dtype=torch.float32)
        P0 = N * H * W

        def attn_bwd(k, xname, dcat, dgated_out, dgated_prev, last):
            pre = f"attn_{k}"
            T = sp.attn[k]
            h, w, hh, wh, Ck = T["dims"]
            D = T["grad"] = {}
            D["dwy"], D["dy"], D["dxg"] = new(h, w, Ck), new(h, w, Ck), new(h, w, Ck)
            D["dsigup"] = torch.empty(N, h, w, 1, dtype=torch.float32, device=device)
            D["dsig"] = torch.empty(N, hh, wh, 1, dtype=torch.float32, device=device)
            D["dpsip"], D["df"], D["dxt"], D["dphi"] = new(hh, wh, 32), new(hh, wh, Ck), new(h, w, Ck), new(Hg, Wg, Ck)
            bn = _mod(self.owner, pre + ".W.1")
            items = [
                ("bn_bwd", (V(T["wy"]), V(dcat), V(D["dwy"]), N * h * w, Ck, pre, T["save"])),
                wg(pre + ".W.0", T["y"], D["dwy"], h, w, Ck, Ck, k=1, pad=0, bias=True),
                cv(V(D["dwy"]), V(D["dy"]), Wp("b", pre + ".W.0"), N, h, w, Ck, Ck, ksize=1, pad=0),
                call(lambda x=V(B[xname]), g=T["sigup"], dy=V(D["dy"]), dx=V(D["dxg"]), dg=D["dsigup"], n=N * h * w, Ck=Ck: A.check(
                    L.srganfd_gate_mul(1, x, g.data_ptr(), dy, dx, dg.data_ptr(), dtc, n, Ck, st()), "gate_mul_bwd")),
                rs(2, V(D["dsigup"]), V(D["dsig"]), hh, wh, 1, A.F32),
                call(lambda ds=D["dsig"], s_=T["sig"]: A.check(L.srganfd_sigmoid_bwd(ds.data_ptr(), s_.data_ptr(), ds.data_ptr(), ds.numel(), st()), "sigmoid_bwd")),
                call(lambda ds=D["dsig"], o=V(D["dpsip"]), hh=hh, wh=wh: A.check(
                    L.srganfd_nchw_to_nhwc(ds.data_ptr(), N, 1, hh, wh, o, dtc, 32, None, None, st()), "pad32")),
                wg(pre + ".psi", T["f"], D["dpsip"], hh, wh, Ck, 32, k=1, pad=0, cout_real=1, bias=True),
                cv(V(D["dpsip"]), V(D["df"]), Wp("b", pre + ".psi"), N, hh, wh, 32, Ck, ksize=1, pad=0, mask=V(T["f"]), mask_slope=0.0),
                wg(pre + ".theta", B[xname], D["df"], h, w, Ck, Ck, k=2, s=2, pad=0),
            ]
            items += strided_dgrad(lambda par: ("b", pre + ".theta", par), 1, D["df"], D["dxt"], hh, wh, Ck, Ck)
            items += [
                call(lambda a=V(D["df"]), b=V(D["dphi"]), hh=hh, wh=wh, Ck=Ck: A.check(
                    L.srganfd_resize_bilinear(1, a, b, dtc, N, Hg, Wg, hh, wh, Ck, st()), "resize_bwd")),
                wg(pre + ".phi", B["gated"], D["dphi"], Hg, Wg, 4 * nf, Ck, k=1, pad=0, bias=True),
                cv(V(D["dphi"]), V(dgated_out), Wp("b", pre + ".phi"), N, Hg, Wg, Ck, 4 * nf, ksize=1, pad=0,
                   r1=V(dgated_prev) if dgated_prev is not None else A.NULL_VIEW, r1_scale=1.0 if dgated_prev is not None else 0.0,
                   mask=V(B["gated"]) if last else A.NULL_VIEW, mask_slope=0.2),
            ]
            return items

        if sp.thin_o:
            head = [("thin", ops.ThinLaunch("thin_wgrad", ops.thin_args(dtc, N, H, W, 1, P("conv9.weight"), V(B["c8"]), w_big_is_cout=False, thin=sp.dl),
                                            dw_off=self._poff("conv9.weight"), db_off=self._poff("conv9.bias"), ws=sp.thin_ws)),
                    ("thin", ops.ThinLaunch("thin_in", ops.thin_args(dtc, N, H, W, 1, P("conv9.weight"), V(G["g8"]), w_big_is_cout=False, flip=True,
                                                                     mask=V(B["c8"]), mask_slope=0.2, thin=sp.dl)))]
        else:
            head = [wg("conv9", B["c8"], sp.dl, H, W, nf, 32, cout_real=1, bias=True),
                    cv(V(sp.dl), V(G["g8"]), Wp("b", "conv9"), N, H, W, 32, nf, mask=V(B["c8"]), mask_slope=0.2)]
        bw: List[tuple] = head + [
            wg("conv8", B["c7"], G["g8"], H, W, nf, nf, sn=True),
            cv(V(G["g8"]), V(G["g7"]), Wp("b", "conv8"), N, H, W, nf, nf, mask=V(B["c7"]), mask_slope=0.2),
            wg("conv7", B["x6"], G["g7"], H, W, nf, nf, sn=True),
            cv(V(G["g7"]), V(G["g6"]), Wp("b", "conv7"), N, H, W, nf, nf, mask=V(B["x6"]), mask_slope=0.2),
            wg("conv6", B["cat3"], G["g6"], H, W, 2 * nf, nf, sn=True),
            cv(V(G["g6"]), V(G["dc3"]), Wp("b", "conv6"), N, H, W, nf, 2 * nf),
            lb(V(G["dc3"], c0=nf), V(B["cat3"], c0=nf), V(G["dc3"], c0=nf), P0, nf),
            wg("cat_3.convU", B["b5"], G["dc3"], H, W, 2 * nf, nf, sn=True, dy_c0=nf),
            cv(V(G["dc3"], c0=nf), V(G["db5"]), Wp("b", "cat_3.convU"), N, H, W, nf, 2 * nf),
            rs(2, V(G["db5"]), V(G["dx5"]), *R[1], 2 * nf),
            lb(V(G["dx5"]), V(B["x5"]), V(G["dx5"]), P0 // 4, 2 * nf),
        ]
        bw += attn_bwd(3, "x0", G["dc3"], G["dgated"][0], None, False)
        bw += [
            wg("conv5", B["cat2"], G["dx5"], *R[1], 4 * nf, 2 * nf, sn=True),
            cv(V(G["dx5"]), V(G["dc2"]), Wp("b", "conv5"), N, *R[1], 2 * nf, 4 * nf),
            lb(V(G["dc2"], c0=2 * nf), V(B["cat2"], c0=2 * nf), V(G["dc2"], c0=2 * nf), P0 // 4, 2 * nf),
            wg("cat_2.convU", B["b4"], G["dc2"], *R[1], 4 * nf, 2 * nf, sn=True, dy_c0=2 * nf),
            cv(V(G["dc2"], c0=2 * nf), V(G["db4"]), Wp("b", "cat_2.convU"), N, *R[1], 2 * nf, 4 * nf),
            rs(2, V(G["db4"]), V(G["dx4"]), *R[2], 4 * nf),
            lb(V(G["dx4"]), V(B["x4"]), V(G["dx4"]), P0 // 16, 4 * nf),
        ]
        bw += attn_bwd(2, "x1", G["dc2"], G["dgated"][1], G["dgated"][0], False)
        bw += [
            wg("conv4", B["cat1"], G["dx4"], *R[2], 8 * nf, 4 * nf, sn=True),
            cv(V(G["dx4"]), V(G["dc1"]), Wp("b", "conv4"), N, *R[2], 4 * nf, 8 * nf),
            lb(V(G["dc1"], c0=4 * nf), V(B["cat1"], c0=4 * nf), V(G["dc1"], c0=4 * nf), P0 // 16, 4 * nf),
            wg("cat_1.convU", B["b3"], G["dc1"], *R[2], 8 * nf, 4 * nf, sn=True, dy_c0=4 * nf),
            cv(V(G["dc1"], c0=4 * nf), V(G["db3"]), Wp("b", "cat_1.convU"), N, *R[2], 4 * nf, 8 * nf),
            rs(2, V(G["db3"]), V(G["dx3a"]), *R[3], 8 * nf),
        ]
        bw += attn_bwd(1, "x2", G["dc1"], G["dgated"][2], G["dgated"][1], True)
        dgated = G["dgated"][2]      # sum of the three gates' gradients, LeakyReLU' of `gated` applied
        bw += [
            wg("gating", B["x3"], dgated, *R[3], 8 * nf, 4 * nf, k=1, pad=1, sn=True),
            cv(V(dgated), V(G["dx3"]), Wp("b", "gating"), N, Hg, Wg, 4 * nf, 8 * nf, ksize=1, pad=-1, r1=V(G["dx3a"]), r1_scale=1.0,
               mask=V(B["x3"]), mask_slope=0.2),
            wg("conv3", B["x2"], G["dx3"], *R[2], 4 * nf, 8 * nf, s=2, sn=True),
        ]
        bw += strided_dgrad(lambda par: ("b", "conv3", par), 2, G["dx3"], G["dx2"], *R[3], 8 * nf, 4 * nf,
                            r1=sp.attn[1]["grad"]["dxg"], r2=sp.attn[1]["grad"]["dxt"], mask=B["x2"])
        bw.append(wg("conv2", B["x1"], G["dx2"], *R[1], 2 * nf, 4 * nf, s=2, sn=True))
        bw += strided_dgrad(lambda par: ("b", "conv2", par), 2, G["dx2"], G["dx1"], *R[2], 4 * nf, 2 * nf,
                            r1=sp.attn[2]["grad"]["dxg"], r2=sp.attn[2]["grad"]["dxt"], mask=B["x1"])
        bw.append(wg("conv1", B["x0"], G["dx1"], *R[0], nf, 2 * nf, s=2, sn=True))
        bw += strided_dgrad(lambda par: ("b", "conv1", par), 2, G["dx1"], G["dx0"], *R[1], 2 * nf, nf,
                            r1=sp.attn[3]["grad"]["dxg"], r2=sp.attn[3]["grad"]["dxt"], mask=B["x0"])
        if sp.thin_i:
            bw.append(("thin", ops.ThinLaunch("thin_wgrad", ops.thin_args(dtc, N, H, W, self.in_ch, P("conv0.weight"), V(G["dx0"]), w_big_is_cout=True, thin=B["xin"]),
                                              dw_off=self._poff("conv0.weight"), db_off=self._poff("conv0.bias"), ws=sp.thin_ws)))
            sp.dx_conv = ops.ThinLaunch("thin_out", ops.thin_args(dtc, N, H, W, self.in_ch, P("conv0.weight"), V(G["dx0"]), w_big_is_cout=True, flip=True,
                                                                  thin_out=sp.dxp, thin_out_pitch=4))
        else:
            bw.append(wg("conv0", B["xin"], G["dx0"], H, W, 32, nf, cin_real=self.in_ch, bias=True))
            sp.dx_conv = ops.conv_args(dtc, V(G["dx0"]), V(sp.dxp), Wp("b", "conv0"), N, H, W, nf, 32, cout_store=self.in_ch, y_f32=True)
        sp.bw = bw
        sp.wg_ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
        sp.gtmp = torch.zeros(self.fp.total, dtype=torch.float32, device=device)
        sp.sn_ws = torch.empty(len(SN_LAYERS) * A.SN_GRAD_WS_FLOATS, dtype=torch.float32, device=device)

    # ---- execution ----
    def _run_conv(self, L, st, a, rec, what):
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
        self._spectral_norm_and_pack(pk, training)
        N, _, H, W = x.shape
        sp = self._plan(N, H, W, dt, dtc, dev, pk)
        L, st = A.lib(), A.stream_ptr()
        x = x.contiguous().float()
        A.check(L.srganfd_nchw_to_nhwc(x.data_ptr(), N, self.in_ch, H, W, A.view(sp.B["xin"]), dtc, sp.B["xin"].shape[-1], None, None, st), "nchw_to_nhwc")
        logits = torch.empty(N, 1, H, W, dtype=torch.float32, device=dev)
        rec = profiling.REC
        for kind, item in sp.fw + [sp.conv9(logits)]:
            if kind == "thin":
                item.launch(rec)
            elif kind == "conv":
                self._run_conv(L, st, item, rec, "conv2d")
            elif kind == "bn":
                k, xv, yv, npix, Ck, gamma, beta, bn, save = item
                if bn.running_mean.device != dev:
                    raise A.SrganfdError("BatchNorm buffers must live on the module's GPU")
                if training and self.sync_bn is not None:
                    # (sum x, sum x^2) of this rank's pixels -> summed over the ranks -> statistics of the whole batch (pixel count = ranks * npix)
                    sb, ws = self.sync_bn, sp.bn_ws
                    args = (xv, yv, dtc, npix, Ck, gamma, beta, bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.momentum, bn.eps,
                            save.data_ptr(), ws.data_ptr(), 1.0)
                    A.check(L.srganfd_batchnorm_fwd_sync(*args, 1, 0, st), "batchnorm_fwd_sync")
                    sb.all_reduce(ws[:L.srganfd_batchnorm_partial_floats(Ck)])
                    A.check(L.srganfd_batchnorm_fwd_sync(*args, 2, npix * sb.world, st), "batchnorm_fwd_sync")
                else:
                    A.check(L.srganfd_batchnorm_fwd(xv, yv, dtc, npix, Ck, gamma, beta, bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                                                    bn.momentum, bn.eps, 1 if training else 0, save.data_ptr(), sp.bn_ws.data_ptr(), st), "batchnorm_fwd")
                if training:
                    bn.num_batches_tracked += 1
            else:
                item()
        o = self.owner
        o.ly1, o.ly2, o.ly3 = (sp.attn[k]["sigup"].view(N, 1, *sp.attn[k]["dims"][:2]).clone() for k in (1, 2, 3))
        self.token += 1
        sp.token = self.token
        sp.inv_sigma = pk["scalars"]
        sp.training = training
        self._last = sp
        return logits

    def backward(self, sp, token, dlogits, need_wgrad, need_dx):
        if getattr(sp, "token", None) != token:
            raise A.SrganfdError("discriminator activations / spectral-norm state were overwritten by a later forward before backward ran")
        if not sp.training:
            raise A.SrganfdError("UNetDiscriminatorAesrgan backward is implemented for training-mode forwards (BatchNorm batch statistics)")
        L, st = A.lib(), A.stream_ptr()
        N, H, W, dtc = sp.N, sp.H, sp.W, sp.dtc
        dlogits = dlogits.contiguous().float()
        A.check(L.srganfd_nchw_to_nhwc(dlogits.data_ptr(), N, 1, H, W, A.view(sp.dl), dtc, sp.dl.shape[-1], None, None, st), "nchw_to_nhwc")
        flat = self.fp.flat
        # the flat gradient also receives BatchNorm's dgamma/dbeta; frozen-parameter passes write them to scratch
        flat_grad = self.fp.new_grad(sp.device) if need_wgrad else sp.gtmp
        rec = profiling.REC
        sn_grads = []
        for item in sp.bw:
            kind = item[0]
            if kind == "conv":
                self._run_conv(L, st, item[1], rec, "conv2d(dgrad)")
            elif kind == "thin":
                if item[1].is_wgrad and not need_wgrad:
                    continue
                item[1].launch(rec, flat_grad.data_ptr())
            elif kind == "wgrad":
                if not need_wgrad:
                    continue
                _, plan, xv, dyv, sn_index, name, ks = item
                dst = flat_grad if sn_index is None else sp.gtmp
                run = lambda: A.check(L.srganfd_conv2d_wgrad(plan.host, plan.dev.data_ptr(), xv, dyv, dst.data_ptr(), None, sp.wg_ws.data_ptr(),
                                                             sp.wg_ws.numel(), st), "conv2d_wgrad")
                if rec is None:
                    run()
                else:
                    rec.bracket(plan.label, (plan.flops, plan.nbytes), run)
                if sn_index is not None:
                    co, ci = self._wshape(name)
                    off = 4 * self._poff(name + ".weight_orig")
                    m = _mod(self.owner, name)
                    sn_grads.append((sp.gtmp.data_ptr() + off, flat.data_ptr() + off, m.weight_u.data_ptr(), m.weight_v.data_ptr(),
                                     sp.inv_sigma.data_ptr() + 4 * (2 * sn_index + 1), flat_grad.data_ptr() + off, co, ci * ks * ks))
            elif kind == "bn_bwd":
                xv, dyv, dxv, npix, Ck, pre, save = item[1]
                if self.sync_bn is not None:
                    # dgamma/dbeta stay this rank's sums (the flat-gradient all-reduce averages them with everything else); the
                    # dx coefficients need (sum dy, sum dy*xhat) over the whole batch
                    sb, ws = self.sync_bn, sp.bn_ws
                    nfl = L.srganfd_batchnorm_partial_floats(Ck)
                    if sp.bn_ws_global is None:
                        sp.bn_ws_global = torch.empty(L.srganfd_batchnorm_partial_floats(256), dtype=torch.float32, device=ws.device)
                    args = (xv, dyv, dxv, dtc, npix, Ck, flat.data_ptr() + 4 * self._poff(pre + ".W.1.weight"), save.data_ptr(),
                            flat_grad.data_ptr() + 4 * self._poff(pre + ".W.1.weight"), flat_grad.data_ptr() + 4 * self._poff(pre + ".W.1.bias"),
                            0.0, ws.data_ptr(), sp.bn_ws_global.data_ptr(), A.NULL_VIEW, 1.0)
                    A.check(L.srganfd_batchnorm_bwd_sync(*args, 1, 0, st), "batchnorm_bwd_sync")
                    sp.bn_ws_global[:nfl].copy_(ws[:nfl])
                    sb.all_reduce(sp.bn_ws_global[:nfl])
                    A.check(L.srganfd_batchnorm_bwd_sync(*args, 2, npix * sb.world, st), "batchnorm_bwd_sync")
                else:
                    A.check(L.srganfd_batchnorm_bwd(xv, dyv, dxv, dtc, npix, Ck, flat.data_ptr() + 4 * self._poff(pre + ".W.1.weight"), save.data_ptr(),
                                                    flat_grad.data_ptr() + 4 * self._poff(pre + ".W.1.weight"),
                                                    flat_grad.data_ptr() + 4 * self._poff(pre + ".W.1.bias"), 0.0, sp.bn_ws.data_ptr(), st), "batchnorm_bwd")
            else:
                item[1]()
        ops.spectral_norm_grad_batch(sn_grads, sp.sn_ws)       # dL/d(W/sigma) -> dL/dW_orig for every normalised layer, batched
        dx = None
        if need_dx:
            if type(sp.dx_conv) is ops.ThinLaunch:
                sp.dx_conv.launch(rec)
            else:
                A.check(L.srganfd_conv2d(C.byref(sp.dx_conv), st), "conv2d(dgrad conv0)")
            dx = torch.empty(N, self.in_ch, H, W, dtype=torch.float32, device=sp.device)
            A.check(L.srganfd_nhwc_to_nchw(A.view(sp.dxp), A.F32, N, self.in_ch, H, W, dx.data_ptr(), 0, st), "nhwc_to_nchw")
        return (flat_grad if need_wgrad else None), dx


class _AesrganFn(torch.autograd.Function):
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


def aesrgan_engine(owner: nn.Module) -> AesrganDiscriminatorEngine:
    return _engine(owner, lambda: AesrganDiscriminatorEngine(owner))


def aesrgan_discriminator_apply(owner: nn.Module, x: Tensor) -> Tensor:
    eng = aesrgan_engine(owner)
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in eng.fp.params)):
        return _AesrganFn.apply(x, eng, owner.training, *eng.fp.params)
    return eng.forward(x, owner.training)
