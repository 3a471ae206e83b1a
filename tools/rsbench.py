"""Timing + output digests of the bilinear x2 resampling entry points at the U-Net discriminator's three up-block shapes (B=32, 512x512
input: 512ch@64^2, 256ch@128^2, 128ch@256^2 low-res).  Run twice (SRGANFD_LIB=<other build>) and compare digests for bit-equality."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sr_gan_fd_amd import _abi as A  # noqa: E402

L = A.lib()
B = int(os.environ.get("B", "32"))
dt, dtc = torch.float16, A.F16


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def digest(t):
    return hashlib.sha1(t.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes()).hexdigest()[:12]


torch.manual_seed(0)
for (c, h) in ((512, 64), (256, 128), (128, 256), (40, 37)):
    n = B if c != 40 else 3
    lo = torch.randn(n, h, h, c, device="cuda").to(dt)
    hi = torch.empty(n, 2 * h, 2 * h, c, device="cuda", dtype=dt)
    st = A.stream_ptr()
    f = lambda: A.check(L.srganfd_resample(1, A.view(lo), A.view(hi), dtc, n, h, h, c, st), "fwd")
    us = timed(f)
    gb = (lo.numel() + hi.numel()) * 2 / 1e9
    print(f"fwd  c={c:4d} h={h:4d}: {us:8.1f} us  {gb / us * 1e6:7.0f} GB/s  {digest(hi)}")
    g = torch.randn(n, 2 * h, 2 * h, c, device="cuda").to(dt)
    act = torch.randn(n, h, h, c, device="cuda").to(dt)
    raw, msk = torch.empty_like(act), torch.empty_like(act)
    f = lambda: A.check(L.srganfd_resample_bwd_lrelu(A.view(g), A.view(raw), A.view(act), A.view(msk), dtc, n, h, h, c, 0.2, st), "bwd")
    us = timed(f)
    gb = (g.numel() + 3 * act.numel()) * 2 / 1e9
    print(f"bwd+ c={c:4d} h={h:4d}: {us:8.1f} us  {gb / us * 1e6:7.0f} GB/s  {digest(raw)} {digest(msk)}")
    f = lambda: A.check(L.srganfd_resample(2, A.view(g), A.view(raw), dtc, n, h, h, c, st), "bwd")
    us = timed(f)
    gb = (g.numel() + act.numel()) * 2 / 1e9
    print(f"bwd  c={c:4d} h={h:4d}: {us:8.1f} us  {gb / us * 1e6:7.0f} GB/s  {digest(raw)}")

# 2x2 reductions: op 0 nearest-x2 adjoint (generator tail), op 3 max pool (VGG-19), op 4 relu copy; (h, w) = the ABI's h, w arguments
for (op, c, h, n) in ((0, 64, 128, B), (0, 64, 256, B), (3, 64, 512, B), (3, 128, 256, B), (3, 256, 128, B), (3, 512, 64, B), (4, 64, 512, B), (0, 24, 9, 3), (3, 24, 18, 3)):
    st = A.stream_ptr()
    ih = 2 * h if op == 0 else h
    oh = h // 2 if op == 3 else h
    src = torch.randn(n, ih, ih, c, device="cuda").to(dt)
    dst = torch.empty(n, oh, oh, c, device="cuda", dtype=dt)
    f = lambda: A.check(L.srganfd_resample(op, A.view(src), A.view(dst), dtc, n, h, h, c, st), "resample")
    us = timed(f)
    gb = (src.numel() + dst.numel()) * 2 / 1e9
    print(f"op{op}  c={c:4d} h={h:4d}: {us:8.1f} us  {gb / us * 1e6:7.0f} GB/s  {digest(dst)}")
