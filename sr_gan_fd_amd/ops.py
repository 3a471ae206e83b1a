"""Thin host-side wrappers over the C ABI: weight packing, fused conv, weight gradients.

Host code is plumbing only (device memory, streams); all arithmetic runs in libsrganfd_hip.so.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import torch

from . import _abi as A

DT = {torch.bfloat16: A.BF16, torch.float32: A.F32, torch.float16: A.F16}


def esize(dtype_code: int) -> int:
    return 4 if dtype_code == A.F32 else 2


def struct_array_to_device(arr, device) -> torch.Tensor:
    """Upload a ctypes array of PODs; returns the uint8 device tensor that owns the bytes."""
    raw = bytes(memoryview(arr).cast("B"))
    return torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)


class PackTable:
    """A device-resident table of pack jobs (offset based, reusable every step)."""

    def __init__(self, jobs: Sequence[A.PackJob], device):
        self.n = len(jobs)
        arr = (A.PackJob * self.n)(*jobs)
        self.max_elems = max(j.ksize * j.ksize * j.k * j.n for j in jobs)
        self.dev = struct_array_to_device(arr, device)

    def run(self, params: torch.Tensor, packed: torch.Tensor, scalars: Optional[torch.Tensor] = None):
        A.check(A.lib().srganfd_pack_weights(self.dev.data_ptr(), self.n, self.max_elems, params.data_ptr(),
                                             scalars.data_ptr() if scalars is not None else None,
                                             packed.data_ptr(), A.stream_ptr()), "pack_weights")


def pack_job(dst_off: int, dtype: int, ksize: int, k: int, n: int, segs: Sequence[dict]) -> A.PackJob:
    j = A.PackJob()
    j.dst_off, j.dtype, j.ksize, j.k, j.n, j.nseg = dst_off, dtype, ksize, k, n, len(segs)
    # B-fragment order of the MFMA form the consuming kernel runs (the library knows its own dispatch: srganfd_set_mfma16)
    j.layout = A.lib().srganfd_pack_layout(dtype, ksize, n)
    assert 1 <= len(segs) <= 5 and dst_off % 16 == 0
    for i, s in enumerate(segs):
        g = j.seg[i]
        g.src_off = s["src_off"]; g.scale_off = s.get("scale_off", -1)
        g.co_src, g.ci_src = s["co_src"], s["ci_src"]
        g.k_lo, g.k_len = s.get("k_lo", 0), s["k_len"]
        g.co_off, g.ci_off = s.get("co_off", 0), s.get("ci_off", 0)
        g.transposed = s.get("transposed", 0); g.scale = s.get("scale", 1.0)
    return j


def packed_bytes(dtype: int, ksize: int, k: int, n: int) -> int:
    return ksize * ksize * k * n * esize(dtype)


def pad32(c: int) -> int:
    return (c + 31) // 32 * 32


def pack_single(weight: torch.Tensor, dtype: int, transposed: bool = False, scale: float = 1.0) -> torch.Tensor:
    """Pack one (Cout, Cin, k, k) fp32 weight for conv2d (forward) or for its data gradient."""
    co, ci, kh, kw = weight.shape
    assert kh == kw
    k, n = (pad32(co), pad32(ci)) if transposed else (pad32(ci), pad32(co))
    out = torch.empty(packed_bytes(dtype, kh, k, n), dtype=torch.uint8, device=weight.device)
    w = weight.detach().contiguous().float()
    job = pack_job(0, dtype, kh, k, n, [dict(src_off=0, co_src=co, ci_src=ci, k_len=k, transposed=int(transposed), scale=scale)])
    PackTable([job], weight.device).run(w, out)
    return out


def conv_args(dtype: int, x: A.View, y: A.View, w_packed, n: int, h_in: int, w_in: int, cin: int, cout: int, *,
              cout_store: Optional[int] = None, ksize: int = 3, stride: int = 1, pad: int = 1, up: int = 0,
              bias=None, act: int = A.ACT_NONE, slope: float = 0.2, alpha: float = 1.0, post_scale: float = 1.0,
              r1: A.View = A.NULL_VIEW, r1_scale: float = 0.0, r2: A.View = A.NULL_VIEW, r2_scale: float = 0.0,
              mask: A.View = A.NULL_VIEW, mask_slope: float = 0.2, alpha_dev=None, y_f32: bool = False,
              y2: A.View = A.NULL_VIEW) -> A.ConvArgs:
    a = A.ConvArgs()
    a.dtype, a.n, a.h_in, a.w_in, a.up = dtype, n, h_in, w_in, up
    a.ksize, a.stride, a.pad, a.cin, a.cout = ksize, stride, pad, cin, cout
    a.cout_store = cout if cout_store is None else cout_store
    hl, wl = h_in << up, w_in << up
    a.h_out, a.w_out = (hl + 2 * pad - ksize) // stride + 1, (wl + 2 * pad - ksize) // stride + 1
    a.x, a.y, a.r1, a.r2, a.mask, a.y2 = x, y, r1, r2, mask, y2
    a.w_packed = w_packed if isinstance(w_packed, int) else w_packed.data_ptr()
    a.bias = None if bias is None else (bias if isinstance(bias, int) else bias.data_ptr())
    a.alpha_dev = None if alpha_dev is None else (alpha_dev if isinstance(alpha_dev, int) else alpha_dev.data_ptr())
    a.alpha, a.slope, a.post_scale, a.r1_scale, a.r2_scale, a.mask_slope = alpha, slope, post_scale, r1_scale, r2_scale, mask_slope
    a.act, a.y_f32 = act, int(y_f32)
    return a


CLASS4_ENABLED = os.environ.get("SRGANFD_CLASS4", "1") != "0"   # same-box A/B switch: 0 launches the four parity classes separately


def class4_ok(dtype: int, n_out: int, offsets: Sequence[int], pack_bytes: int, ksize: int = 2) -> bool:
    """can the four output-parity classes of a stride-2 data gradient go out as ONE launch (srganfd_conv_args.out_classes = 4)?
    16-bit modes, 2x2- or 1x1-tap classes, the output's channel blocks (64 wide for the 2x2 kernel over a multiple of 64 channels,
    else 32) a power of two, the four packed operands back to back."""
    if n_out % 32:
        return False
    nb = n_out // (64 if ksize == 2 and n_out % 64 == 0 else 32)
    return (CLASS4_ENABLED and dtype != A.F32 and ksize in (1, 2) and nb > 0 and nb & (nb - 1) == 0
            and all(offsets[c] == offsets[0] + c * pack_bytes for c in range(4)))


def conv2d(args: A.ConvArgs) -> None:
    A.check(A.lib().srganfd_conv2d(C.byref(args), A.stream_ptr()), "conv2d")


# ---- LDS-resident dense-block launch (csrc/dense_chain.hip): the five convs of a dense block, or of its data-gradient pass, as one launch ----
DENSE_CHAIN = os.environ.get("SRGANFD_DENSE_CHAIN", "auto")     # "0": never, "1": whenever the arguments qualify, "auto": small launches only (see dense_chain_wanted)
_DC_WS = {}


def dense_chain_workspace(device) -> torch.Tensor:
    """hand-off flags + error word of the dense-chain launches of one device (zeroed once here; the launches re-zero the flags they use
    on the stream, and they run in stream order, so every chain of a device can share it)"""
    ws = _DC_WS.get(str(device))
    if ws is None:
        ws = _DC_WS[str(device)] = torch.zeros(int(A.lib().srganfd_dense_chain_workspace_bytes()), dtype=torch.uint8, device=device)
    return ws


def dense_chain_giveups(device):
    """hand-off waits that gave up since the device's workspace was allocated (None: no dense-chain launch was ever planned there);
    synchronises -- for the end of a run, not for the step"""
    ws = _DC_WS.get(str(device))
    return None if ws is None else int(ws[:4].view(torch.int32).item())


def dense_chain_wanted(n: int, h: int, w: int, cus: int = 256) -> bool:
    """the launch needs every 16 x 16 tile of a pass resident at once (one workgroup per CU, cus // tiles images per pass) and keeps
    one flag per growth layer and tile of the call.  "auto" takes it where it measured faster than the five launches
    (profiles/r05_dense_chain_v2_bench.txt): batches that fit ONE pass -- 1.5x (forward) / 1.35x (data gradient) at the reference's
    crop sizes; a second pass costs a whole pass whatever it holds (batch 16 at 72 x 72: 400 tiles, 0.96x / 0.83x) and at batch 32,
    128 x 128 (eight passes) the per-layer launches are 1.2x faster"""
    if DENSE_CHAIN == "0":
        return False
    per = -(-h // 16) * -(-w // 16)
    if per > cus or n * per > 16384:
        return False
    return DENSE_CHAIN == "1" or n * per <= cus


class DenseChain:
    """n conv launches (A.ConvArgs, the arguments srganfd_conv2d would get) run by srganfd_dense_chain; ``ok`` False when the library
    refuses them (then the caller keeps the separate launches)."""

    def __init__(self, layers: Sequence[A.ConvArgs], device):
        self.n = len(layers)
        self.arr = (A.ConvArgs * self.n)()
        for i, a in enumerate(layers):
            C.memmove(C.byref(self.arr, i * C.sizeof(A.ConvArgs)), C.byref(a), C.sizeof(A.ConvArgs))
        self.layers = list(layers)           # keeps the structs (and the label cache on them) alive
        self.ok = A.lib().srganfd_dense_chain_check(self.arr, self.n) == 0
        self.ws = dense_chain_workspace(device) if self.ok else None
        self.flops = sum(2.0 * a.n * a.h_out * a.w_out * 9 * a.cin * a.cout for a in layers)
        # algorithmic bytes of the launch: the block input read once, every layer's output written once, the epilogue operands read once
        es, px = 2, layers[0].n * layers[0].h_out * layers[0].w_out
        self.bytes = float(px * es * (layers[0].cin + sum(a.cout * (1 + bool(a.r1.ptr) + bool(a.r2.ptr) + bool(a.mask.ptr)) for a in layers)))
        self.label = "dense_chain_kernel<%s,%d layers%s>" % (A.DT_NAME[layers[0].dtype], self.n, ",mask" if layers[0].mask.ptr else "")
        self.work = (self.flops, self.bytes)

    def run(self) -> None:
        A.check(A.lib().srganfd_dense_chain(self.arr, self.n, self.ws.data_ptr(), self.ws.numel(), A.stream_ptr()), "dense_chain")

    def launch(self, rec=None) -> None:
        if rec is None:
            self.run()
        else:
            rec.bracket(self.label, self.work, self.run)

    def errors(self) -> int:
        """hand-off waits that gave up since the workspace was allocated (synchronises; tests)"""
        return int(self.ws[:4].view(torch.int32).item())


def dense_chain_or_launches(layers: Sequence[A.ConvArgs], device) -> list:
    """[DenseChain] when the batch is in the chain's regime and the library accepts the launches, else the launches themselves"""
    a = layers[0]
    if a.dtype in (A.F16, A.BF16) and dense_chain_wanted(a.n, a.h_in, a.w_in):
        ch = DenseChain(layers, device)
        if ch.ok:
            return [ch]
    return list(layers)


# ---- thin-side convolutions (csrc/conv_thin.hip): 1..4 channels against 64, 3x3 stride 1 pad 1, 16-bit dtypes ----
THIN_ENABLED = os.environ.get("SRGANFD_THIN", "1") != "0"      # same-box A/B switch: 0 keeps these layers on the 32-channel-padded kernels


def thin_ok(dtype: int, big_channels: int, thin_channels: int, ksize: int = 3) -> bool:
    """does this layer run on the thin kernels?  (16-bit modes, 64 channels against 1..4, 3x3: everything else -- and the exact-fp32
    parity mode -- stays on srganfd_conv2d with the thin side padded to 32 channels)"""
    return THIN_ENABLED and dtype != A.F32 and big_channels == 64 and 1 <= thin_channels <= 4 and ksize == 3


def _ptr(t):
    return None if t is None else (t if isinstance(t, int) else t.data_ptr())


def thin_args(dtype: int, n: int, h: int, w: int, cs: int, weight, big: A.View, *, w_big_is_cout: bool, flip: bool = False, bias=None,
              act: int = A.ACT_NONE, slope: float = 0.2, mask: A.View = A.NULL_VIEW, mask_slope: float = 0.2, thin=None, thin_out=None,
              thin_out_pitch: int = 4) -> A.ThinArgs:
    """weight: the layer's raw fp32 parameter (Cout, Cin, 3, 3) -- tensor or device address; thin: NHWC4 16-bit tensor / address"""
    a = A.ThinArgs()
    a.dtype, a.n, a.h, a.w, a.cs = dtype, n, h, w, cs
    a.w_big_is_cout, a.flip, a.act, a.slope, a.mask_slope = int(w_big_is_cout), int(flip), act, slope, mask_slope
    a.weight, a.bias, a.big, a.mask = _ptr(weight), _ptr(bias), big, mask
    a.thin, a.thin_out, a.thin_out_pitch = _ptr(thin), _ptr(thin_out), thin_out_pitch
    return a


def thin_work(a: A.ThinArgs, kind: str):
    """(algorithmic FLOP, algorithmic bytes) of one thin launch: 2 * pixels * 9 * 64 * cs; 64-channel tensor (+ mask) once, thin tensor once"""
    px = float(a.n) * a.h * a.w
    nbytes = px * (128.0 + (128.0 if a.mask.ptr else 0.0) + (4.0 * a.thin_out_pitch if kind == "thin_out" else 8.0))
    return 2.0 * px * 9 * 64 * a.cs, nbytes


def thin_in(a: A.ThinArgs) -> None:
    A.check(A.lib().srganfd_conv2d_thin_in(C.byref(a), A.stream_ptr()), "conv2d_thin_in")


def thin_out(a: A.ThinArgs) -> None:
    A.check(A.lib().srganfd_conv2d_thin_out(C.byref(a), A.stream_ptr()), "conv2d_thin_out")


def thin_wgrad_workspace_bytes() -> int:
    return int(A.lib().srganfd_conv2d_thin_wgrad_workspace())


def thin_wgrad(a: A.ThinArgs, dw, db, workspace: torch.Tensor) -> None:
    A.check(A.lib().srganfd_conv2d_thin_wgrad(C.byref(a), _ptr(dw), _ptr(db), workspace.data_ptr(), workspace.numel() * workspace.element_size(),
                                              A.stream_ptr()), "conv2d_thin_wgrad")


class ThinLaunch:
    """One thin-side launch in an engine's launch list.  kind: "thin_in" / "thin_out" / "thin_wgrad" (the latter writes the weight and
    bias gradient at element offsets ``dw_off`` / ``db_off`` of the flat gradient whose address ``launch`` is given)."""

    def __init__(self, kind: str, args: A.ThinArgs, dw_off: int = -1, db_off: int = -1, ws: Optional[torch.Tensor] = None, keep=()):
        self.kind, self.args, self.dw_off, self.db_off, self.ws = kind, args, dw_off, db_off, ws
        self.keep = keep                  # tensors whose addresses the struct holds
        self.is_wgrad = kind == "thin_wgrad"
        self.label = "%s_kernel<%s%s>" % (kind, A.DT_NAME[args.dtype], ",mask" if args.mask.ptr else "")
        self.work = thin_work(args, kind)

    def run(self, grad_ptr: int = 0) -> None:
        L, st, a = A.lib(), A.stream_ptr(), self.args
        if self.kind == "thin_in":
            rc = L.srganfd_conv2d_thin_in(C.byref(a), st)
        elif self.kind == "thin_out":
            rc = L.srganfd_conv2d_thin_out(C.byref(a), st)
        else:
            rc = L.srganfd_conv2d_thin_wgrad(C.byref(a), grad_ptr + 4 * self.dw_off, (grad_ptr + 4 * self.db_off) if self.db_off >= 0 else None,
                                             self.ws.data_ptr(), self.ws.numel() * self.ws.element_size(), st)
        if rc:
            A.check(rc, self.kind)

    def launch(self, rec=None, grad_ptr: int = 0) -> None:
        if rec is None:
            self.run(grad_ptr)
        else:
            rec.bracket(self.label, self.work, lambda: self.run(grad_ptr))


class WgradPlan:
    """Host+device plan of one weight-gradient launch (several convs sharing x and dy)."""

    def __init__(self, device, dtype: int, n: int, h_in: int, w_in: int, x_channels: int, dy_channels: int,
                 convs: Sequence[dict], ksize: int = 3, stride: int = 1, pad: int = 1, up: int = 0, splits: int = 0):
        hl, wl = h_in << up, w_in << up
        s = A.WgradShape(dtype, n, h_in, w_in, up, ksize, stride, pad, (hl + 2 * pad - ksize) // stride + 1,
                         (wl + 2 * pad - ksize) // stride + 1, x_channels, dy_channels, len(convs), splits)
        carr = (A.WgradConv * len(convs))()
        for i, c in enumerate(convs):
            w = carr[i]
            w.ci_lo, w.cin, w.co_lo, w.cout = c.get("ci_lo", 0), c["cin"], c.get("co_lo", 0), c["cout"]
            w.dw_off, w.db_off = c["dw_off"], c.get("db_off", -1)
            w.co_dst, w.ci_dst = c["co_dst"], c["ci_dst"]
            w.alpha, w.beta, w.alpha_off = c.get("alpha", 1.0), c.get("beta", 0.0), c.get("alpha_off", -1)
        ho, wo = s.h_out, s.w_out
        self.flops = sum(2.0 * n * ho * wo * ksize * ksize * c["co_dst"] * c["ci_dst"] for c in convs)
        es = 4 if dtype == A.F32 else 2
        # algorithmic bytes: x and dy read once, fp32 gradients written once
        self.nbytes = float(n * h_in * w_in * x_channels * es + n * ho * wo * dy_channels * es
                            + sum(4.0 * ksize * ksize * c["co_dst"] * c["ci_dst"] for c in convs))
        self.label = f"wgrad_kernel<{A.DT_NAME[dtype]},KS={ksize},S={stride}>+reduce"
        L = A.lib()
        nbytes = L.srganfd_wgrad_plan_bytes(C.byref(s), carr)
        if nbytes == 0:
            raise A.SrganfdError("wgrad plan: " + L.srganfd_last_error().decode())
        self.host = C.create_string_buffer(nbytes)
        ws = C.c_size_t(0)
        A.check(L.srganfd_wgrad_plan_build(C.byref(s), carr, self.host, nbytes, C.byref(ws)), "wgrad_plan_build")
        self.workspace_bytes = ws.value
        self.dev = torch.frombuffer(bytearray(self.host.raw), dtype=torch.uint8).to(device)

    def run(self, x: A.View, dy: A.View, grads: torch.Tensor, workspace: torch.Tensor,
            scalars: Optional[torch.Tensor] = None) -> None:
        assert workspace.numel() * workspace.element_size() >= self.workspace_bytes
        A.check(A.lib().srganfd_conv2d_wgrad(self.host, self.dev.data_ptr(), x, dy, grads.data_ptr(),
                                             scalars.data_ptr() if scalars is not None else None,
                                             workspace.data_ptr(), workspace.numel() * workspace.element_size(),
                                             A.stream_ptr()), "conv2d_wgrad")


# same-box A/B switch (0: one group of launches per layer, the round-1 form); both forms run in the HIP library and agree to the bit
_SN_BATCH = os.environ.get("SRGANFD_SN_BATCH", "1") != "0"


def spectral_norm_batch(layers: Sequence[tuple], training: bool, workspace: torch.Tensor, eps: float = 1e-12) -> None:
    """layers: (w_orig_ptr, u_ptr, v_ptr, rows, cols, sigma_ptr, inv_sigma_ptr); all of them in ceil(n / 8) x 4 launches."""
    if not _SN_BATCH:
        for (w, u, v, rows, cols, sig, isig) in layers:
            A.check(A.lib().srganfd_spectral_norm(w, u, v, rows, cols, 1 if training else 0, eps, sig, isig, workspace.data_ptr(), A.stream_ptr()), "spectral_norm")
        return
    arr = (A.SnJob * len(layers))()
    off, base = 0, workspace.data_ptr()
    for q, (w, u, v, rows, cols, sig, isig) in zip(arr, layers):
        q.w_orig, q.u, q.v, q.sigma_out, q.inv_sigma_out, q.rows, q.cols = w, u, v, sig, isig, rows, cols
        q.workspace = base + 4 * off
        off += A.sn_ws_floats(rows, cols)
    assert off <= workspace.numel() and workspace.dtype == torch.float32
    A.check(A.lib().srganfd_spectral_norm_batch(arr, len(layers), 1 if training else 0, eps, A.stream_ptr()), "spectral_norm_batch")


def spectral_norm_grad_batch(layers: Sequence[tuple], workspace: torch.Tensor, beta: float = 0.0) -> None:
    """layers: (g_weight_ptr, w_orig_ptr, u_ptr, v_ptr, inv_sigma_ptr, dw_orig_ptr, rows, cols)."""
    if not layers:
        return
    if not _SN_BATCH:
        for (g, w, u, v, isig, dw, rows, cols) in layers:
            A.check(A.lib().srganfd_spectral_norm_grad(g, w, u, v, isig, dw, rows, cols, beta, workspace.data_ptr(), A.stream_ptr()), "spectral_norm_grad")
        return
    arr = (A.SnGradJob * len(layers))()
    assert len(layers) * A.SN_GRAD_WS_FLOATS <= workspace.numel() and workspace.dtype == torch.float32
    for i, (q, (g, w, u, v, isig, dw, rows, cols)) in enumerate(zip(arr, layers)):
        q.g_weight, q.w_orig, q.u, q.v, q.inv_sigma, q.dw_orig, q.rows, q.cols = g, w, u, v, isig, dw, rows, cols
        q.workspace = workspace.data_ptr() + 4 * i * A.SN_GRAD_WS_FLOATS
    A.check(A.lib().srganfd_spectral_norm_grad_batch(arr, len(layers), beta, A.stream_ptr()), "spectral_norm_grad_batch")
