"""Eager vs hipGraph replay of the generator-only iteration at small batches.  python tools/graph_bench.py"""
import sys, time, torch
sys.path.insert(0, '.')
from sr_gan_fd_amd import model as M
from sr_gan_fd_amd.trainer import GeneratorTrainer
from sr_gan_fd_amd.graph import GraphedStep

def run(b, h, nrrdb, dtype, graphed, steps=10):
    torch.manual_seed(0)
    g = M.bsrgan_x4(num_rrdb=nrrdb); g.compute_dtype = dtype; g.cuda().train()
    tr = GeneratorTrainer(g, lr=1e-4, betas=(0.9, 0.99), eps=1e-4, ema_decay=0.999)
    lr, gt = torch.rand(b, 3, h, h, device='cuda'), torch.rand(b, 3, 4 * h, 4 * h, device='cuda')
    step = GraphedStep(tr, lr, gt) if graphed else tr.step
    for _ in range(3): step(lr, gt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step(lr, gt)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

cfgs = ((32, 128, torch.bfloat16),) if '--big' in sys.argv else ((4, 32, torch.bfloat16), (4, 32, torch.float32), (1, 64, torch.bfloat16), (8, 64, torch.bfloat16))
for (b, h, dt) in cfgs:
    e, g = run(b, h, 23, dt, False), run(b, h, 23, dt, True)
    print(f"G-only 23 RRDB B={b} {h}->{4*h} {str(dt)[6:]}: eager {e:7.2f} ms/step ({b/e*1e3:7.1f} img/s)   graph replay {g:7.2f} ms/step ({b/g*1e3:7.1f} img/s)")
