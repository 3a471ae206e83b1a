"""conv_rows (row-stream conv) against conv_igemm on the same arguments: results and time per launch."""
import sys, os
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

dt, dtc = torch.float16, A.F16
torch.manual_seed(0)
N = int(os.environ.get("RB_N", 32))
cases = [  # (name, H, cin, cout, planar, epilogue)
    ("64->32 planar lrelu", 128, 64, 32, 1, "lrelu"), ("96->32 planar lrelu", 128, 96, 32, 1, "lrelu"), ("128->32 planar lrelu", 128, 128, 32, 1, "lrelu"),
    ("160->32 planar lrelu", 128, 160, 32, 1, "lrelu"), ("160->32 planar mask", 128, 160, 32, 1, "mask"), ("64->32 planar mask", 128, 64, 32, 1, "mask"),
    ("192->64 planar r1", 128, 192, 64, 1, "r1"),
    ("64->64 nhwc lrelu 512", 512, 64, 64, 0, "lrelu"), ("64->64 nhwc mask 512", 512, 64, 64, 0, "mask"), ("128->64 nhwc r1+y2 512", 512, 128, 64, 0, "r1y2"),
    ("64->128 nhwc 512", 512, 64, 128, 0, "none"), ("128->128 nhwc relu 256", 256, 128, 128, 0, "relu"),
]
for name, H, cin, cout, planar, epi in cases:
    W_ = H
    Cbuf = 192 if planar else cin
    x = (torch.randn(N, H, W_, Cbuf, device="cuda") * 0.5).to(dt)       # planar buffers hold the same bytes: interpret per view
    Cy = 192 if planar else cout
    w = torch.randn(cout, cin, 3, 3, device="cuda") * (1.0 / (3 * cin ** 0.5))
    b = torch.randn(cout, device="cuda") * 0.1
    aux = (torch.randn(N, H, W_, Cy, device="cuda")).to(dt)
    outs = {}
    times = {}
    for lay in (None, 2):
        wp = ops.pack_single(w, dtc, layout=lay)
        y = torch.zeros(N, H, W_, Cy, dtype=dt, device="cuda")
        y2 = torch.zeros(N, H, W_, Cy, dtype=dt, device="cuda")
        xv = A.view(x, c0=0, planar=planar)
        yv = A.view(y, c0=(64 if (planar and cout == 32) else 0), planar=planar)
        av = A.view(aux, c0=(64 if (planar and cout == 32) else 0), planar=planar)
        kw = dict(w_layout=2) if lay == 2 else {}
        if epi == "lrelu": kw.update(bias=b, act=A.ACT_LRELU, slope=0.2)
        elif epi == "relu": kw.update(bias=b, act=A.ACT_RELU)
        elif epi == "mask": kw.update(mask=av, mask_slope=0.2)
        elif epi == "r1": kw.update(bias=b, post_scale=0.2, r1=av, r1_scale=1.0)
        elif epi == "r1y2": kw.update(act=A.ACT_LRELU, slope=0.2, r1=av, r1_scale=1.0, y2=A.view(y2, c0=0, planar=planar))
        a = ops.conv_args(dtc, xv, yv, wp, N, H, W_, cin, cout, **kw)
        ops.conv2d(a)
        torch.cuda.synchronize()
        outs[lay] = (y.clone(), y2.clone())
        times[lay] = timeit(lambda: ops.conv2d(a))
    ref, got = outs[None], outs[2]
    err = ((got[0].float() - ref[0].float()).abs().max() / (ref[0].float().abs().max() + 1e-30)).item()
    err2 = ((got[1].float() - ref[1].float()).abs().max() / (ref[1].float().abs().max() + 1e-30)).item()
    gf = 2.0 * N * H * W_ * 9 * cin * cout / 1e9
    print("%-26s igemm %7.1f us (%6.1f TF/s)   rows %7.1f us (%6.1f TF/s)  ratio %.3f   max rel diff %.2e / y2 %.2e" %
          (name, times[None], gf / times[None] * 1e-3 * 1e3 / 1e3 * 1e3, times[2], gf / times[2] * 1e-3 * 1e3 / 1e3 * 1e3, times[2] / times[None], err, err2))
