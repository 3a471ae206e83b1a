"""EXPERIMENT: bilinear x2 adjoint + LeakyReLU' (srganfd_resample_bwd_lrelu) variants at the discriminator's three shapes, batch 32"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sr_gan_fd_amd import _abi as A
L = A.lib()
N = 32
dt = torch.float16
for (h, c) in ((256, 128), (128, 256), (64, 512)):
    dy = torch.randn(N, 2 * h, 2 * h, c, device="cuda", dtype=dt)
    act = torch.randn(N, h, h, c, device="cuda", dtype=dt)
    raw = torch.empty_like(act); msk = torch.empty_like(act)
    ref = None
    for var in (0, 1, 2, 3, 4, 5, 0, 1, 2, 3, 4, 5):   # variants selected by SRGANFD_BILBWD in the experiment build of this file (git history); the product keeps 4 rows + nt above 192 MB
        os.environ["SRGANFD_BILBWD"] = str(var)
        call = lambda: A.check(L.srganfd_resample_bwd_lrelu(A.view(dy), A.view(raw), A.view(act), A.view(msk), A.F16, N, h, h, c, 0.2, A.stream_ptr()), "x")
        for _ in range(3): call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): call()
        e1.record(); torch.cuda.synchronize()
        if ref is None: ref = (raw.clone(), msk.clone())
        ok = torch.equal(ref[0], raw) and torch.equal(ref[1], msk)
        by = (dy.numel() + 3 * act.numel()) * 2
        us = e0.elapsed_time(e1) * 100
        print(f"h={h} c={c} var={var} {us:8.1f} us  {by / us / 1e6:6.2f} TB/s  equal={ok}", flush=True)
