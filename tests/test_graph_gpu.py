"""hipGraph replay of the fused iterations == the eager iterations (same seeds, same batches)."""
import time

import numpy as np
import pytest
import torch

from tests.util import scaled_init

pytestmark = pytest.mark.gpu


def _gen(dtype):
    from sr_gan_fd_amd import model as M
    torch.manual_seed(0)
    g = M.bsrgan_x4(in_channels=3, out_channels=3, channels=64, growth_channels=32, num_rrdb=3)
    scaled_init(g, 3.0, 0.5)
    g.compute_dtype = dtype
    return g.cuda().train()


def _batches(n, b=4, h=32):
    gen = torch.Generator(device="cuda").manual_seed(5)
    return [(torch.rand(b, 3, h, h, device="cuda", generator=gen), torch.rand(b, 3, 4 * h, 4 * h, device="cuda", generator=gen)) for _ in range(n)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_graphed_generator_step_equals_eager(dtype):
    from sr_gan_fd_amd.graph import GraphedStep
    from sr_gan_fd_amd.trainer import GeneratorTrainer
    data = _batches(6)
    ge, gg = _gen(dtype), _gen(dtype)
    te = GeneratorTrainer(ge, lr=1e-4, betas=(0.9, 0.99), eps=1e-4, ema_decay=0.999)
    tg = GeneratorTrainer(gg, lr=1e-4, betas=(0.9, 0.99), eps=1e-4, ema_decay=0.999)
    for _ in range(2):                       # what GraphedStep's warm-up does, on the same example batch
        te.step(*data[0])
    step = GraphedStep(tg, *data[0], warmup=2)
    losses_e, losses_g = [], []
    for lr, gt in data[1:]:
        losses_e.append(te.step(lr, gt).item())
        losses_g.append(step(lr, gt).item())
    print("eager", losses_e, "graphed", losses_g)
    assert np.allclose(losses_e, losses_g, rtol=1e-6, atol=0)
    a, b = te.flat, tg.flat
    assert ((a - b).abs().max() / a.abs().max()).item() < 1e-6     # bias corrections: device pow vs host pow, else identical kernels
    assert ((te.opt.ema - tg.opt.ema).abs().max() / te.opt.ema.abs().max()).item() < 1e-6
    assert tg.opt.t == te.opt.t and int(tg.opt.step_dev.item()) == te.opt.t


def test_graphed_gan_step_equals_eager():
    from sr_gan_fd_amd import model as M
    from sr_gan_fd_amd.gan import GanTrainer
    from sr_gan_fd_amd.graph import GraphedStep
    data = _batches(5, b=2, h=16)

    def build():
        g = _gen(torch.float32)
        torch.manual_seed(1)
        d = M.discriminator_unet(in_channels=3, out_channels=1, channels=64)
        d.compute_dtype = torch.float32
        return GanTrainer(g, d.cuda().train(), None)
    te, tg = build(), build()
    for _ in range(2):
        te.step(*data[0])
    step = GraphedStep(tg, *data[0], warmup=2)
    for lr, gt in data[1:]:
        se = te.step(lr, gt).cpu().numpy().copy()
        sg = step(lr, gt).cpu().numpy().copy()
        assert np.allclose(se, sg, rtol=1e-5, atol=1e-7), (se, sg)
    for oe, og in ((te.g_opt, tg.g_opt), (te.d_opt, tg.d_opt)):
        assert ((oe.flat - og.flat).abs().max() / oe.flat.abs().max()).item() < 1e-5
    # timing is reported, not asserted (tools/graph_bench.py measures the launch-bound configurations)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        te.step(*data[1])
    torch.cuda.synchronize()
    t_e = (time.perf_counter() - t0) / 5
    t0 = time.perf_counter()
    for _ in range(5):
        step(*data[1])
    torch.cuda.synchronize()
    t_g = (time.perf_counter() - t0) / 5
    print(f"GAN step B=2 16->64: eager {t_e * 1e3:.2f} ms, graph replay {t_g * 1e3:.2f} ms")


def test_graphed_f16_step_handles_overflow_like_eager():
    """f16 trainers carry the dynamic loss scale (trainer.LossScaler) in device memory: the captured kernels read the scale, write
    found_inf and apply GradScaler.update() at every replay exactly as the eager step does (no re-capture, no host read-back).
    With the scale forced far too high (2^37: the gradient seeds overflow f16 for several halvings) both runs skip the same steps, back off the same
    way -- same scale after every iteration -- and end with the same parameters."""
    from sr_gan_fd_amd.graph import GraphedStep
    from sr_gan_fd_amd.trainer import GeneratorTrainer
    data = _batches(9)
    te = GeneratorTrainer(_gen(torch.float16), lr=1e-4, betas=(0.9, 0.99), eps=1e-4, ema_decay=0.999)
    tg = GeneratorTrainer(_gen(torch.float16), lr=1e-4, betas=(0.9, 0.99), eps=1e-4, ema_decay=0.999)
    for _ in range(2):
        te.step(*data[0])
    step = GraphedStep(tg, *data[0], warmup=2)
    te.step(*data[1])
    step(*data[1])                                       # one clean replayed step first
    te.scaler.report(), tg.scaler.report()               # fold everything in, then force the overflow
    te.scaler.scale = tg.scaler.scale = 2.0 ** 37
    w_before = tg.flat.clone()
    scales = []
    for lr, gt in data[2:]:
        te.step(lr, gt)
        step(lr, gt)
        scales.append((te.scaler.scale, tg.scaler.scale))
    torch.cuda.synchronize()
    re, rg = te.scaler.report(), tg.scaler.report()
    print("eager", re, "graphed", rg, scales)
    assert re == rg and re["skipped"] >= 3 and re["scale"] < 2.0 ** 37
    assert all(a == b for a, b in scales)
    assert torch.isfinite(tg.flat).all()
    assert ((te.flat - tg.flat).abs().max() / te.flat.abs().max()).item() < 1e-6
    if re["skipped"] == len(data) - 2:
        assert torch.equal(tg.flat, w_before)            # every forced step overflowed: the parameters never moved
