"""EXPERIMENT (upper bound): stream-level concurrency at the reference's own crop size, where a conv launch is one partial round of tiles.
One generator-only trainer of batch B on one stream / two independent trainers of batch B/2 back to back on one stream / the same two on two
HIP streams.  The dense-block launch is switched off (one chain launch at a time per device, include/srganfd.h).
    python tools/r5/two_stream_small.py [B] [lr size]"""
import os, sys, time
os.environ["SRGANFD_DENSE_CHAIN"] = "0"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sr_gan_fd_amd import model as M
from sr_gan_fd_amd.trainer import GeneratorTrainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H = int(sys.argv[2]) if len(sys.argv) > 2 else 72
dev = torch.device("cuda", 0)


def make(b):
    torch.manual_seed(0)
    g = M.bsrgan_x4().to(dev); g.compute_dtype = torch.float16; g.train()
    t = GeneratorTrainer(g, lr=1e-4, betas=(0.9, 0.99), eps=1e-4, ema_decay=0.999)
    return t, torch.rand(b, 3, H, H, device=dev), torch.rand(b, 3, 4 * H, 4 * H, device=dev)


def timed(fn, steps=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e3


t1, lr1, gt1 = make(B)
a, lra, gta = make(B // 2); b, lrb, gtb = make(B // 2)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def serial():
    a.step(lra, gta); b.step(lrb, gtb)


def conc():
    with torch.cuda.stream(sa): a.step(lra, gta)
    with torch.cuda.stream(sb): b.step(lrb, gtb)


for rep in range(2):
    ms = timed(lambda: t1.step(lr1, gt1))
    print("batch %d %d->%d  one stream, one trainer:            %.2f ms/step  %.1f img/s" % (B, H, 4 * H, ms, B * 1e3 / ms), flush=True)
    ms = timed(serial)
    print("batch %d %d->%d  one stream, 2 x batch %d serial:     %.2f ms/pair  %.1f img/s" % (B, H, 4 * H, B // 2, ms, B * 1e3 / ms), flush=True)
    ms = timed(conc)
    print("batch %d %d->%d  two streams, 2 x batch %d:           %.2f ms/pair  %.1f img/s" % (B, H, 4 * H, B // 2, ms, B * 1e3 / ms), flush=True)
