"""Time the thin-side kernels against the 32-channel-padded conv_igemm / wgrad launches they replace (B=32, 512x512, f16)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sr_gan_fd_amd import _abi as A, ops

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

N, H, W = int(os.environ.get("TB_N", 32)), 512, 512
dt, dtc = torch.float16, A.F16
big = torch.randn(N, H, W, 64, device="cuda").to(dt)
act = torch.randn(N, H, W, 64, device="cuda").to(dt)
thin = torch.zeros(N, H, W, 4, dtype=dt, device="cuda"); thin[..., :3] = torch.randn(N, H, W, 3, device="cuda").to(dt)
pad32 = torch.zeros(N, H, W, 32, dtype=dt, device="cuda"); pad32[..., :3] = thin[..., :3]
px = N * H * W
for cs in (3, 1):
    Win = torch.randn(64, cs, 3, 3, device="cuda") * 0.1
    Wout = torch.randn(cs, 64, 3, 3, device="cuda") * 0.1
    b64, bcs = torch.randn(64, device="cuda"), torch.randn(cs, device="cuda")
    y = torch.empty(N, H, W, 64, dtype=dt, device="cuda")
    o4 = torch.empty(N, H, W, 4, dtype=torch.float32, device="cuda")
    ws = torch.empty(ops.thin_wgrad_workspace_bytes(), dtype=torch.uint8, device="cuda")
    a_in = ops.thin_args(dtc, N, H, W, cs, Win, A.view(y), w_big_is_cout=True, bias=b64, act=A.ACT_LRELU, thin=thin)
    a_inm = ops.thin_args(dtc, N, H, W, cs, Wout, A.view(y), w_big_is_cout=False, flip=True, mask=A.view(act), thin=thin)
    a_out = ops.thin_args(dtc, N, H, W, cs, Wout, A.view(big), w_big_is_cout=False, bias=bcs, thin_out=o4)
    a_wg1 = ops.thin_args(dtc, N, H, W, cs, Win, A.view(big), w_big_is_cout=True, thin=thin)
    a_wg0 = ops.thin_args(dtc, N, H, W, cs, Wout, A.view(big), w_big_is_cout=False, thin=thin)
    dw1, db1 = torch.empty_like(Win), torch.empty(64, device="cuda")
    dw0, db0 = torch.empty_like(Wout), torch.empty(cs, device="cuda")
    res = {
        "thin_in": (timeit(lambda: ops.thin_in(a_in)), px * (8 + 128)),
        "thin_in+mask": (timeit(lambda: ops.thin_in(a_inm)), px * (8 + 256)),
        "thin_out": (timeit(lambda: ops.thin_out(a_out)), px * (128 + 16)),
        "thin_wgrad(big=dy)": (timeit(lambda: ops.thin_wgrad(a_wg1, dw1, db1, ws)), px * (128 + 8)),
        "thin_wgrad(big=x)": (timeit(lambda: ops.thin_wgrad(a_wg0, dw0, db0, ws)), px * (128 + 8)),
    }
    # the padded launches they replace
    wp_f = ops.pack_single(torch.nn.functional.pad(Win, (0, 0, 0, 0, 0, 0, 0, 0)), dtc)
    c_in = ops.conv_args(dtc, A.view(pad32), A.view(y), wp_f, N, H, W, 32, 64, bias=b64, act=A.ACT_LRELU)
    wp_b = ops.pack_single(Wout, dtc, transposed=True)
    c_inm = ops.conv_args(dtc, A.view(pad32), A.view(y), wp_b, N, H, W, 32, 64, mask=A.view(act), mask_slope=0.2)
    wp_o = ops.pack_single(Wout, dtc)
    c_out = ops.conv_args(dtc, A.view(big), A.view(o4), wp_o, N, H, W, 64, 32, cout_store=cs, bias=bcs, y_f32=True)
    res["padded 32->64"] = (timeit(lambda: ops.conv2d(c_in)), px * (64 + 128))
    res["padded 32->64+mask"] = (timeit(lambda: ops.conv2d(c_inm)), px * (64 + 256))
    res["padded 64->%d" % cs] = (timeit(lambda: ops.conv2d(c_out)), px * (128 + 16))
    g = torch.empty(64 * 32 * 9 + 64, device="cuda")
    p1 = ops.WgradPlan("cuda", dtc, N, H, W, 32, 64, [dict(cin=32, cout=64, dw_off=0, db_off=64 * cs * 9, co_dst=64, ci_dst=cs)])
    wsb = torch.empty(p1.workspace_bytes, dtype=torch.uint8, device="cuda")
    res["padded wgrad %d->64" % cs] = (timeit(lambda: p1.run(A.view(pad32), A.view(big), g, wsb)), px * (64 + 128))
    p0 = ops.WgradPlan("cuda", dtc, N, H, W, 64, 32, [dict(cin=64, cout=32, dw_off=0, db_off=64 * cs * 9, co_dst=cs, ci_dst=64)])
    wsb0 = torch.empty(p0.workspace_bytes, dtype=torch.uint8, device="cuda")
    res["padded wgrad 64->%d" % cs] = (timeit(lambda: p0.run(A.view(big), A.view(pad32), g, wsb0)), px * (128 + 64))
    print("thin channels %d, N=%d %dx%d" % (cs, N, H, W))
    for k, (us, nb) in res.items():
        print("  %-24s %8.1f us  %7.1f GB/s (bytes the launch must move: %.0f MB)" % (k, us, nb / us / 1e3, nb / 1e6))
