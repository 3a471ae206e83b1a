"""EXPERIMENT: is the small-shape generator-only iteration bound by the host's launch rate once the weight gradients go to a side stream?
Host time to ENQUEUE one iteration (no synchronisation inside) against the synchronised time per iteration, batch 16, 32 -> 128.
    SRGANFD_WGRAD_STREAM=... python tools/r5/host_time.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sr_gan_fd_amd import model as M          # noqa: E402
from sr_gan_fd_amd.trainer import GeneratorTrainer   # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
if os.environ.get("SRGANFD_BENCH_OWN_STREAM") == "1":
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
B, S = int(os.environ.get("B", "16")), int(os.environ.get("S", "32"))
torch.manual_seed(0)
g = M.bsrgan_x4(in_channels=3, out_channels=3, channels=64, growth_channels=32, num_rrdb=23)
g.compute_dtype = torch.float16
g.to(dev).train()
tr = GeneratorTrainer(g, lr=1e-4, betas=(0.9, 0.99), eps=1e-4, ema_decay=0.999)
x, y = torch.rand(B, 3, S, S, device=dev), torch.rand(B, 3, 4 * S, 4 * S, device=dev)
for _ in range(5):
    tr.step(x, y)
torch.cuda.synchronize()
n = 20
host = 0.0
t0 = time.perf_counter()
for _ in range(n):
    h0 = time.perf_counter()
    tr.step(x, y)
    host += time.perf_counter() - h0
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print("batch %d %dx%d  WGRAD_STREAM=%s CUMASK=%s: host enqueue %.2f ms / iteration, synchronised %.2f ms / iteration" % (
    B, S, S, os.environ.get("SRGANFD_WGRAD_STREAM", "0"), "yes" if os.environ.get("SRGANFD_WGRAD_CUMASK") else "no", 1e3 * host / n, 1e3 * tot / n))
