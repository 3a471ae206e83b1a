"""EXPERIMENT: eager vs hipGraph replay (sr_gan_fd_amd.graph.GraphedStep) of the generator-only iteration at the reference's default shapes (f16)"""
import sys, time, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sr_gan_fd_amd import model as M
from sr_gan_fd_amd.trainer import GeneratorTrainer
from sr_gan_fd_amd.graph import GraphedStep

def run(b, h, graphed, steps=20):
    torch.manual_seed(0)
    g = M.bsrgan_x4(num_rrdb=23); g.cuda().train()
    tr = GeneratorTrainer(g, lr=1e-4, betas=(0.9, 0.99), eps=1e-4, ema_decay=0.999)
    lr, gt = torch.rand(b, 3, h, h, device='cuda'), torch.rand(b, 3, 4 * h, 4 * h, device='cuda')
    step = GraphedStep(tr, lr, gt) if graphed else tr.step
    for _ in range(5): step(lr, gt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step(lr, gt)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

for (b, h) in ((16, 32), (16, 48), (16, 72), (4, 32)):
    e, g = run(b, h, False), run(b, h, True)
    print(f"G-only 23 RRDB batch {b} {h}->{4*h} f16, SRGANFD_DENSE_CHAIN={os.environ.get('SRGANFD_DENSE_CHAIN', 'auto')}: eager {e:7.2f} ms/step ({b/e*1e3:7.1f} img/s)   graph replay {g:7.2f} ms/step ({b/g*1e3:7.1f} img/s)", flush=True)
