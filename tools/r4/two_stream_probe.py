"""EXPERIMENT: what does stream-level concurrency buy?  Two independent generator-only trainers of batch 16 on two HIP streams
(the launch boundaries / pipeline fill and drain of one stream's kernels overlap the other stream's kernels) against one trainer of
batch 32 on one stream, and two batch-16 trainers back to back on ONE stream (the cost of the smaller launches alone)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sr_gan_fd_amd import model as M
from sr_gan_fd_amd.trainer import GeneratorTrainer

dev = torch.device("cuda", 0)
def make(B):
    torch.manual_seed(0)
    g = M.bsrgan_x4().to(dev); g.compute_dtype = torch.float16; g.train()
    t = GeneratorTrainer(g, lr=1e-4, betas=(0.9, 0.99), eps=1e-4, ema_decay=0.999)
    return t, torch.rand(B, 3, 128, 128, device=dev), torch.rand(B, 3, 512, 512, device=dev)

def timed(fn, steps=8, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e3

t32, lr32, gt32 = make(32)
ms = timed(lambda: t32.step(lr32, gt32))
print("one stream, batch 32:            %.2f ms/step  %.1f img/s" % (ms, 32e3 / ms), flush=True)
del t32, lr32, gt32; torch.cuda.empty_cache()
a, lra, gta = make(16); b, lrb, gtb = make(16)
def serial():
    a.step(lra, gta); b.step(lrb, gtb)
ms = timed(serial)
print("one stream, 2 x batch 16 serial: %.2f ms/pair  %.1f img/s" % (ms, 32e3 / ms), flush=True)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def conc():
    with torch.cuda.stream(sa): a.step(lra, gta)
    with torch.cuda.stream(sb): b.step(lrb, gtb)
ms = timed(conc)
print("two streams, 2 x batch 16:       %.2f ms/pair  %.1f img/s" % (ms, 32e3 / ms), flush=True)
