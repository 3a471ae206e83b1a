"""Differentiable single-node VGG-19 content loss: ESRGAN/model.py:258-292.

ESRGAN's ContentLoss is ``F.l1_loss(vgg(sr)[node], vgg(gt)[node])`` with ONE node (``features.34`` in esrgan_config) and,
unlike BSRGAN's five-node version, it stays in the autograd graph: the generator receives its gradient, so the frozen
VGG needs a backward pass.  SR and GT run through the extractor as one 2N batch; every ReLU output of the SR half is kept.
Backward = sign(sr_f - gt_f) / numel at the node, then per conv the data-gradient launch of the implicit-GEMM kernel with
the previous ReLU's mask in its epilogue, max-pool backward fused with the ReLU derivative in front of the pool, and the
final relayout that undoes the normalisation's 1/std.  No weight gradients (the extractor is frozen, model.py:275-277).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch
from torch import Tensor, nn

from . import _abi as A
from . import ops
from . import profiling
from .engine import FlatParams, _dt, _engine, _require_gpu, _Shape, PlanCache


class ContentLossGradEngine:
    def __init__(self, owner: nn.Module):
        self.owner = owner
        self.last = int(owner.feature_model_extractor_nodes[0].split(".")[1])
        feats = owner.features
        if not isinstance(feats[self.last], nn.Conv2d):
            raise A.SrganfdError("differentiable ContentLoss: the node must be a conv of vgg19.features (esrgan_config uses features.34)")
        self.convs = [(i, feats[i]) for i in range(self.last + 1) if isinstance(feats[i], nn.Conv2d)]
        self.fp = FlatParams([(f"features.{i}.{k}", getattr(m, k)) for i, m in self.convs for k in ("weight", "bias")])
        self.shapes = PlanCache()
        self.packed: Dict[int, dict] = {}
        self.token = 0

    def _ensure_packed(self, dtc, device):
        flat = self.fp.sync(device)
        pk = self.packed.get(dtc)
        if pk is None or pk["buf"].device != device or pk.get("flat_ptr") != flat.data_ptr():
            jobs, offs, cur = [], {}, 0
            for i, m in self.convs:
                co, ci = m.weight.shape[:2]
                cip = ops.pad32(ci)
                src = self.fp.off(f"features.{i}.weight")
                for key, k, n, seg in ((("f", i), cip, co, dict(src_off=src, co_src=co, ci_src=ci, k_len=cip)),
                                       (("b", i), co, cip, dict(src_off=src, co_src=co, ci_src=ci, k_len=co, transposed=1))):
                    offs[key] = cur
                    jobs.append(ops.pack_job(cur, dtc, 3, k, n, [seg]))
                    cur += (ops.packed_bytes(dtc, 3, k, n) + 255) // 256 * 256
            pk = dict(table=ops.PackTable(jobs, device), offs=offs, buf=torch.empty(cur, dtype=torch.uint8, device=device), flat_ptr=flat.data_ptr())
            self.packed[dtc] = pk
        if self.fp.stale(pk):
            pk["table"].run(flat, pk["buf"])
        return pk

    def _plan(self, N, H, W, dt, dtc, dev, pk):
        key = (N, H, W, dtc, str(dev), pk["buf"].data_ptr(), self.fp.flat.data_ptr())
        sp = self.shapes.get(key)
        if sp is not None:
            return sp
        sp = _Shape()
        sp.N, sp.H, sp.W, sp.dt, sp.dtc, sp.device = N, H, W, dt, dtc, dev
        es = torch.empty(0, dtype=dt).element_size()
        fptr, wptr, O = self.fp.flat.data_ptr(), pk["buf"].data_ptr(), pk["offs"]
        # features.0 (3 -> 64) and its data gradient on the thin-side kernels in the 16-bit modes (csrc/conv_thin.hip): 4-channel pitch
        sp.thin = ops.thin_ok(dtc, self.owner.features[0].weight.shape[0], 3)
        sp.xin = torch.empty(2 * N, H, W, 4 if sp.thin else 32, dtype=dt, device=dev)
        half = lambda t: A.View(t.data_ptr(), t.shape[3], 0)                     # SR half (first N images)
        other = lambda t: A.View(t.data_ptr() + t[:N].numel() * es, t.shape[3], 0)   # GT half
        fw, chain = [], []          # chain: (kind, index, tensor, h, w, c) in forward order
        cur, ch, h, w = sp.xin, 32, H, W
        sp.keep = [sp.xin]
        for idx in range(self.last + 1):
            m = self.owner.features[idx]
            if isinstance(m, nn.Conv2d):
                co = m.weight.shape[0]
                out = torch.empty(2 * N, h, w, co, dtype=dt, device=dev)
                if idx == 0 and sp.thin:
                    fw.append(("thin", ops.ThinLaunch("thin_in", ops.thin_args(dtc, 2 * N, h, w, 3, fptr + 4 * self.fp.off("features.0.weight"), A.view(out),
                                                                               w_big_is_cout=True, bias=fptr + 4 * self.fp.off("features.0.bias"),
                                                                               act=A.ACT_NONE if idx == self.last else A.ACT_RELU, thin=sp.xin))))
                else:
                    fw.append(("conv", ops.conv_args(dtc, A.view(cur), A.view(out), wptr + O[("f", idx)], 2 * N, h, w, ch, co,
                                                     bias=fptr + 4 * self.fp.off(f"features.{idx}.bias"),
                                                     act=A.ACT_NONE if idx == self.last else A.ACT_RELU)))
                chain.append(("conv", idx, cur, out, h, w, ch, co))
                cur, ch = out, co
            elif isinstance(m, nn.MaxPool2d):
                out = torch.empty(2 * N, h // 2, w // 2, ch, dtype=dt, device=dev)
                fw.append(("pool", (A.view(cur), A.view(out), h, w, ch)))
                chain.append(("pool", idx, cur, out, h, w, ch, ch))
                cur, h, w = out, h // 2, w // 2
            sp.keep.append(cur)
        sp.fw = fw
        sp.feat, sp.feat_dims = cur, (h, w, ch)
        sp.loss_views = (half(cur), other(cur))
        sp.ws = torch.empty(A.LOSS_WS_FLOATS, dtype=torch.float32, device=dev)
        # ---- backward launch list (SR half only) ----
        bw = []
        g = torch.empty(N, h, w, ch, dtype=dt, device=dev)
        sp.g_tap = g
        sp.keep.append(g)
        sp.dxp = torch.empty(N, H, W, 4, dtype=torch.float32, device=dev)
        for j in range(len(chain) - 1, -1, -1):
            kind, idx, tin, tout, hh, ww, cin, cout = chain[j]
            if kind == "conv":
                if idx == 0:
                    if sp.thin:
                        bw.append(("thin", ops.ThinLaunch("thin_out", ops.thin_args(dtc, N, hh, ww, 3, fptr + 4 * self.fp.off("features.0.weight"), A.view(g),
                                                                                    w_big_is_cout=True, flip=True, thin_out=sp.dxp, thin_out_pitch=4))))
                    else:
                        bw.append(("conv", ops.conv_args(dtc, A.view(g), A.view(sp.dxp), wptr + O[("b", 0)], N, hh, ww, cout, 32, cout_store=3, y_f32=True)))
                    break
                prev_kind = chain[j - 1][0]
                gin = torch.empty(N, hh, ww, cin, dtype=dt, device=dev)
                sp.keep.append(gin)
                # the conv's input is a ReLU output (mask it here) or a pooled map (the pool's backward applies the ReLU')
                mask = half(tin) if prev_kind == "conv" else A.NULL_VIEW
                bw.append(("conv", ops.conv_args(dtc, A.view(g), A.view(gin), wptr + O[("b", idx)], N, hh, ww, cout, cin, mask=mask, mask_slope=0.0)))
                g = gin
            else:
                gin = torch.empty(N, hh, ww, cin, dtype=dt, device=dev)
                sp.keep.append(gin)
                bw.append(("poolbwd", (half(tin), A.view(g), A.view(gin), hh, ww, cin)))
                g = gin
        sp.bw = bw
        self.shapes[key] = sp
        return sp

    def forward(self, sr: Tensor, gt: Tensor) -> Tensor:
        _require_gpu(sr)
        dt, dtc = _dt(self.owner)
        dev = sr.device
        pk = self._ensure_packed(dtc, dev)
        N, Cin, H, W = sr.shape
        if Cin != 3 or H % 16 or W % 16:
            raise A.SrganfdError("ContentLoss needs 3-channel inputs with height/width multiples of 16 (four 2x2 max-pools)")
        sp = self._plan(N, H, W, dt, dtc, dev, pk)
        L, st = A.lib(), A.stream_ptr()
        mean, std = self.owner.mean, self.owner.std
        for img, hf in ((sr, 0), (gt, 1)):
            img = img.detach().contiguous().float()
            cpad = sp.xin.shape[-1]
            dst = A.View(sp.xin.data_ptr() + hf * N * H * W * cpad * sp.xin.element_size(), cpad, 0)
            A.check(L.srganfd_nchw_to_nhwc(img.data_ptr(), N, 3, H, W, dst, dtc, cpad, mean.data_ptr(), std.data_ptr(), st), "nchw_to_nhwc")
        rec = profiling.REC
        for kind, item in sp.fw:
            if kind == "thin":
                item.launch(rec)
            elif kind == "conv":
                if rec is None:
                    A.check(L.srganfd_conv2d(C.byref(item), st), "conv2d(vgg)")
                else:
                    rec.bracket(profiling.conv_label(item), profiling.conv_work(item), lambda: A.check(L.srganfd_conv2d(C.byref(item), st), "conv2d(vgg)"))
            else:
                xv, yv, h, w, c = item
                A.check(L.srganfd_resample(3, xv, yv, dtc, 2 * N, h, w, c, st), "maxpool")
        loss = torch.zeros(1, dtype=torch.float32, device=dev)
        h, w, c = sp.feat_dims
        A.check(L.srganfd_l1_loss_views(sp.loss_views[0], sp.loss_views[1], dtc, N * h * w, c, 0, 1.0, loss.data_ptr(), 0, sp.ws.data_ptr(), st), "l1_views")
        self.token += 1
        sp.token = self.token
        self._last = sp
        return loss.view(())

    def backward(self, sp, token, dloss: Optional[Tensor], weight: float = 1.0, upstream_ptr: Optional[int] = None) -> Tensor:
        """d(weight * loss)/d(sr) times the upstream scalar: ``dloss`` (autograd's 1-element tensor) or, for the fused trainers,
        ``upstream_ptr`` -- the address of a device float such as the loss scale -- or nothing (1)."""
        if getattr(sp, "token", None) != token:
            raise A.SrganfdError("VGG activations were overwritten by a later ContentLoss forward before backward ran")
        L, st = A.lib(), A.stream_ptr()
        N, dtc = sp.N, sp.dtc
        h, w, c = sp.feat_dims
        if dloss is not None:
            dloss = dloss.detach().contiguous().float()
            upstream_ptr = dloss.data_ptr()
        A.check(L.srganfd_l1_grad_views(sp.loss_views[0], sp.loss_views[1], A.view(sp.g_tap), dtc, N * h * w, c, upstream_ptr,
                                        weight / float(N * h * w * c), st), "l1_grad_views")
        rec = profiling.REC
        for kind, item in sp.bw:
            if kind == "thin":
                item.launch(rec)
            elif kind == "conv":
                if rec is None:
                    A.check(L.srganfd_conv2d(C.byref(item), st), "conv2d(vgg dgrad)")
                else:
                    rec.bracket(profiling.conv_label(item), profiling.conv_work(item), lambda: A.check(L.srganfd_conv2d(C.byref(item), st), "conv2d(vgg dgrad)"))
            else:
                xv, dyv, dxv, hh, ww, cc = item
                A.check(L.srganfd_maxpool2_relu_bwd(xv, dyv, dxv, dtc, N, hh, ww, cc, st), "maxpool2_relu_bwd")
        dsr = torch.empty(N, 3, sp.H, sp.W, dtype=torch.float32, device=sp.device)
        A.check(L.srganfd_nhwc_to_nchw_scaled(A.view(sp.dxp), N, 3, sp.H, sp.W, dsr.data_ptr(), self.owner.std.data_ptr(), st), "nhwc_to_nchw_scaled")
        return dsr


class _ContentFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sr, gt, eng):
        loss = eng.forward(sr, gt)
        ctx.eng, ctx.sp, ctx.token = eng, eng._last, eng.token
        return loss

    @staticmethod
    def backward(ctx, dloss):
        return ctx.eng.backward(ctx.sp, ctx.token, dloss), None, None


def content_loss_single_apply(owner: nn.Module, sr: Tensor, gt: Tensor) -> Tensor:
    eng = _engine(owner, lambda: ContentLossGradEngine(owner))
    if torch.is_grad_enabled() and sr.requires_grad:
        return _ContentFn.apply(sr, gt, eng)
    return eng.forward(sr, gt)
