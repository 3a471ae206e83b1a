"""Micro-benchmark single fused-conv launches, every kernel choice interleaved in ONE process (cdna guide rule 24).

    python tools/kbench.py [--modes 0,1,2,3] [--rounds 5] [--reps 20] [--dtype bf16|f16] [--nhwc]

mode 0 = conv_igemm tiles, 1.. = conv3x3_ring configurations (srganfd_set_ring_mode).  Shapes = the dense-block launches of
BASELINE configs[1] (B=32, 128x128, planar 192-channel buffers) plus the 64->64 tail conv at 512x512."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sr_gan_fd_amd import _abi as A, ops, profiling


def _switch(name, v):
    """A/B switches exist in -DSRGANFD_EXPERIMENT builds (tools/build_variant.sh + SRGANFD_LIB); the product library runs mode 8 only"""
    f = getattr(A.lib(), name, None)
    if f is not None:
        f(v)
    elif (name, v) not in (("srganfd_set_mfma16", 3), ("srganfd_set_ring_mode", 0), ("srganfd_set_ring_mode", -1)):
        raise SystemExit("kbench: this library has no %s (product build): use --modes 8, or SRGANFD_LIB=<experiment build>" % name)


def arg(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


PLANAR = 0 if "--nhwc" in sys.argv else 1
MODES = [int(m) for m in arg("--modes", "0,1,2,3").split(",")]
ROUNDS, REPS = int(arg("--rounds", "5")), int(arg("--reps", "20"))
DBGS = [int(d) for d in arg("--dbg", "0").split(",")]      # SRGANFD_EXPERIMENT builds only (SRGANFD_LIB=build_exp/libsrganfd_exp.so)
IGV = [int(d) for d in arg("--igv", "0").split(",")]       # experiment builds: srganfd_set_igemm_variant values (bit 8 = run-time epilogue everywhere, low byte = tile variant)
ONLY = arg("--only", "")                                    # substring filter on the shape names
DT = {"bf16": torch.bfloat16, "f16": torch.float16}[arg("--dtype", "bf16")]


def make(name, n, h, w, cin, cout, xC, x0, yC, y0, mask=False, res=False, up=0, planar=None):
    global PLANAR
    keep_planar = PLANAR
    if planar is not None:
        PLANAR = planar
    try:
        return _make(name, n, h, w, cin, cout, xC, x0, yC, y0, mask, res, up)
    finally:
        PLANAR = keep_planar


def _make(name, n, h, w, cin, cout, xC, x0, yC, y0, mask, res, up):
    dtc = ops.DT[DT]
    x = (torch.randn(n, h, w, xC, device='cuda') * (0.0 if "--zeros" in sys.argv else 0.5)).to(DT)
    y = torch.empty(n, h << up, w << up, yC, device='cuda', dtype=DT)
    wt = torch.randn(cout, cin, 3, 3, device='cuda') * (0.0 if "--zeros" in sys.argv else 0.05)
    EXP = hasattr(A.lib(), "srganfd_set_mfma16")
    if EXP:
        A.lib().srganfd_set_mfma16(0)
    wp = ops.pack_single(wt, dtc)
    _switch("srganfd_set_mfma16", 3)
    wp16 = ops.pack_single(wt, dtc)             # the same weights in the 16x16x32 B-fragment order where the kernel wants it (mode 8)
    if EXP:
        A.lib().srganfd_set_mfma16(0)
    kw, keep = {}, [x, y, wp, wp16]
    if mask:
        m = torch.randn(n, h, w, yC, device='cuda').to(DT); keep.append(m)
        kw.update(mask=A.view(m, c0=y0, planar=PLANAR))
    if res:
        kw.update(r1=A.view(x, c0=0, planar=PLANAR), r1_scale=1.0, post_scale=0.2)
    a = ops.conv_args(dtc, A.view(x, c0=x0, planar=PLANAR), A.view(y, c0=y0, planar=PLANAR), wp, n, h, w, cin, cout, act=A.ACT_NONE if res else A.ACT_LRELU, up=up, **kw)
    a16 = ops.conv_args(dtc, A.view(x, c0=x0, planar=PLANAR), A.view(y, c0=y0, planar=PLANAR), wp16, n, h, w, cin, cout, act=A.ACT_NONE if res else A.ACT_LRELU, up=up, **kw)
    return name, (a, a16), keep, 2.0 * n * (h << up) * (w << up) * 9 * cin * cout


def time_one(a, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.conv2d(a)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


if __name__ == "__main__":
    N = int(arg("--batch", "32"))
    if arg("--set", "rdb") == "gan":      # tail / discriminator / VGG-19 shapes of the GAN step (NHWC buffers), batch 8
        D = lambda nm, n, h, cin, cout, **k: make(nm, n, h, h, cin, cout, cin, 0, cout, 0, planar=0, **k)
        shapes = [D("tail up-conv 64->64 128^2 -> 256^2", 8, 128, 64, 64, up=1), D("tail up-conv 64->64 256^2 -> 512^2", 8, 256, 64, 64, up=1),
                  D("64->64 512^2", 8, 512, 64, 64), D("128->64 512^2 (D up3)", 8, 512, 128, 64), D("256->128 256^2 (D up2)", 8, 256, 256, 128),
                  D("512->256 128^2 (D up1)", 8, 128, 512, 256), D("128->128 256^2 (VGG)", 16, 256, 128, 128), D("256->256 128^2 (VGG)", 16, 128, 256, 256),
                  D("512->512 64^2 (VGG)", 16, 64, 512, 512), D("512->512 32^2 (VGG)", 16, 32, 512, 512), D("64->128 256^2 (VGG)", 16, 256, 64, 128)]
    else:
        shapes = [make(f"fwd cin={cin} cout=32", N, 128, 128, cin, 32, 192, 0, 192, cin) for cin in (64, 96, 128, 160)]
    if arg("--set", "rdb") != "gan":
      shapes.append(make("dgrad-like cin=192 cout=32 + mask", N, 128, 128, 192, 32, 192, 0, 192, 160, mask=True))
      shapes.append(make("conv5 cin=192 cout=64 + residual", N, 128, 128, 192, 64, 192, 0, 192, 0, res=True))
      shapes.append(make("tail cin=64 cout=64 512^2 N=8", 8, 512, 512, 64, 64, 64, 0, 64, 0))
    L = A.lib()
    shapes = [s for s in shapes if ONLY in s[0]]
    res = {}
    for rnd in range(ROUNDS + 1):                      # round 0 = warm-up
        for name, (a32, a16), keep, fl in shapes:
            for m in MODES:
                for d, gv in [(d_, g_) for d_ in DBGS for g_ in IGV]:
                    a = a16 if m == 8 else a32           # mode 8: conv_igemm on v_mfma_f32_16x16x32
                    _switch("srganfd_set_mfma16", 3 if m == 8 else 0)
                    _switch("srganfd_set_ring_mode", 0 if m == 8 else m)
                    if len(IGV) > 1 or gv:
                        L.srganfd_set_igemm_variant(gv)       # (after the ring-mode switch, which resets the tile variant)
                    if d or len(DBGS) > 1:
                        L.srganfd_set_debug(d)
                    if hasattr(a, "_kernel_label"):
                        del a._kernel_label
                    lab = profiling.conv_label(a)
                    us = time_one(a, 3 if rnd == 0 else REPS)
                    if rnd:
                        res.setdefault((name, m, d + 100000 * gv, lab, fl), []).append(us)
    _switch("srganfd_set_ring_mode", -1)
    _switch("srganfd_set_mfma16", 3)
    for (name, m, d, lab, fl), v in res.items():
        v.sort()
        med = v[len(v) // 2]
        print(f"{name:38s} mode {m} dbg {d % 100000:2d} igv {d // 100000:3d}: median {med:7.1f} us  min {v[0]:7.1f}  {fl / med / 1e6:7.1f} TFLOP/s   {lab}")
