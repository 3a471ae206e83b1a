"""Micro-benchmark single fused-conv launches (layout / shape experiments).  python tools/kbench.py [--quick|--batch|--chain] [--dbg N] [--planar]"""
import sys, time, torch
sys.path.insert(0, '.')
from sr_gan_fd_amd import _abi as A, ops

PLANAR = 1 if "--planar" in sys.argv else 0     # operand buffers as planar 32-channel groups (srganfd_view.planar)


def run(name, n, h, w, cin, cout, xC, x0, yC, y0, reps=30, mask=False, dt=torch.bfloat16):
    dtc = ops.DT[dt]
    x = torch.randn(n, h, w, xC, device='cuda').to(dt)
    y = torch.empty(n, h, w, yC, device='cuda', dtype=dt)
    wt = torch.randn(cout, cin, 3, 3, device='cuda') * 0.05
    wp = ops.pack_single(wt, dtc)
    kw = {}
    if mask:
        m = torch.randn(n, h, w, yC, device='cuda').to(dt)
        kw = dict(mask=A.view(m, c0=y0, planar=PLANAR))
    a = ops.conv_args(dtc, A.view(x, c0=x0, planar=PLANAR), A.view(y, c0=y0, planar=PLANAR), wp, n, h, w, cin, cout, act=A.ACT_LRELU, **kw)
    for _ in range(3): ops.conv2d(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ops.conv2d(a)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    fl = 2.0 * n * h * w * 9 * cin * cout
    print(f"{name:44s} {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s")

if __name__ == "__main__":
    N = 32
    if "--dbg" in sys.argv:
        A.lib().srganfd_set_debug(int(sys.argv[sys.argv.index("--dbg") + 1]))
    if "--chain" in sys.argv:        # one tile per CU: the latency chain of a single workgroup
        run("N=8 (1 tile/CU) cin=160 cout=32", 8, 128, 128, 160, 32, 192, 0, 192, 160)
        run("N=16 (2 tiles/CU) cin=160 cout=32", 16, 128, 128, 160, 32, 192, 0, 192, 160)
        sys.exit(0)
    if "--batch" in sys.argv:
        for n in (4, 8, 16, 32, 64):
            run(f"N={n} fwd cin=160 cout=32 concat buffer", n, 128, 128, 160, 32, 192, 0, 192, 160)
            run(f"N={n} cin=192 cout=64 x:192 y:192(next)", n, 128, 128, 192, 64, 192, 0, 192, 0)
        sys.exit(0)
    if "--quick" in sys.argv:
        for cin in (64, 96, 128, 160):
            run(f"fwd cin={cin} cout=32 concat buffer", N, 128, 128, cin, 32, 192, 0, 192, 64 + (cin - 64))
        run("dgrad-like cin=192 cout=32 + mask", N, 128, 128, 192, 32, 192, 0, 192, 160, mask=True)
        run("cin=192 cout=64 x:192 y:192(next)", N, 128, 128, 192, 64, 192, 0, 192, 0)
        run("cin=64 cout=64 dense 512^2 N=8", 8, 512, 512, 64, 64, 64, 0, 64, 0)
        sys.exit(0)
    for cin in (64, 128, 160):
        run(f"cin={cin} cout=32 x:192ch y:192ch(slice)", N, 128, 128, cin, 32, 192, 0, 192, 160)
        run(f"cin={cin} cout=32 x:dense y:dense", N, 128, 128, cin, 32, cin, 0, 32, 0)
        run(f"cin={cin} cout=32 x:192ch y:dense", N, 128, 128, cin, 32, 192, 0, 32, 0)
        run(f"cin={cin} cout=32 x:dense y:192ch", N, 128, 128, cin, 32, cin, 0, 192, 160)
    run("cin=192 cout=64 x:192 y:192(next)", N, 128, 128, 192, 64, 192, 0, 192, 0)
    run("cin=192 cout=64 x:192 y:dense", N, 128, 128, 192, 64, 192, 0, 64, 0)
    run("cin=64 cout=64 dense (tail-like, 128^2)", N, 128, 128, 64, 64, 64, 0, 64, 0)
    run("cin=64 cout=64 dense 512^2 N=2", 2, 512, 512, 64, 64, 64, 0, 64, 0)
    # L2-resident repeat: small problem
    run("cin=128 cout=32 N=2 (L2-resident)", 2, 128, 128, 128, 32, 192, 0, 192, 160, reps=100)
