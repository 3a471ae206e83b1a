"""Time the resampling kernels at the discriminator's shapes.  python tools/rsbench.py"""
import sys, torch
sys.path.insert(0, '.')
from sr_gan_fd_amd import _abi as A
L = A.lib()
def t(op, n, h, w, c, reps=10):
    x = torch.randn(n, h, w, c, device='cuda').bfloat16()
    y = torch.empty(n, 2 * h, 2 * w, c, device='cuda', dtype=torch.bfloat16)
    a, b = (x, y) if op == 1 else (y, x)
    for _ in range(2): A.check(L.srganfd_resample(op, A.view(a), A.view(b), A.BF16, n, h, w, c, A.stream_ptr()))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): A.check(L.srganfd_resample(op, A.view(a), A.view(b), A.BF16, n, h, w, c, A.stream_ptr()))
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    gb = (x.numel() + y.numel()) * 2 / 1e9
    print(f"op {op} n={n} {h}x{w}x{c}: {us:8.1f} us  {gb / us * 1e6:6.0f} GB/s  ptrs {x.data_ptr() % 256} {y.data_ptr() % 256}")
for op in (1, 2):
    for (h, c) in ((64, 512), (128, 256), (256, 128)):
        t(op, 32, h, h, c)
