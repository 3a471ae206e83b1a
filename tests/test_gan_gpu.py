"""GPU parity of the discriminator, spectral norm, losses, optimizer, content loss and the full GAN
iteration (through the C ABI) against golden vectors captured from the reference and the CPU oracle.

f32 mode tolerance 1e-3 (north_star); f16 mode (the benchmark dtype, loss-scaled backward) is held to the reference's vectors
too (bounds written next to each assertion); bf16 mode is bounded at 5e-2 and reported.
"""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.util import pinned_vgg, rounded_weights, checksum, load_golden, scaled_init, table, sd_to_params

pytestmark = pytest.mark.gpu
TOL = {torch.float32: 1e-3, torch.bfloat16: 5e-2, torch.float16: 3e-3}


def _rel_l2(a, b):
    a = a.detach().double().cpu() if torch.is_tensor(a) else torch.as_tensor(np.asarray(a)).double()
    b = b.detach().double().cpu() if torch.is_tensor(b) else torch.as_tensor(np.asarray(b)).double()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _rel(a, b):
    a = a.detach().double().cpu() if torch.is_tensor(a) else torch.as_tensor(np.asarray(a)).double()
    b = b.detach().double().cpu() if torch.is_tensor(b) else torch.as_tensor(np.asarray(b)).double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


@pytest.mark.parametrize("op", ["bilinear_fwd", "bilinear_bwd", "nearest_bwd", "maxpool"])
def test_resample_kernels(op):
    from sr_gan_fd_amd import _abi as A
    torch.manual_seed(0)
    n, h, w, c = 2, 6, 10, 32
    L, st = A.lib(), A.stream_ptr()
    if op == "bilinear_fwd":
        x = torch.randn(n, c, h, w)
        ref = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
        a, b = x.permute(0, 2, 3, 1).contiguous().cuda(), torch.empty(n, 2 * h, 2 * w, c, device="cuda")
        A.check(L.srganfd_resample(1, A.view(a), A.view(b), A.F32, n, h, w, c, st))
    elif op in ("bilinear_bwd", "nearest_bwd"):
        x = torch.randn(n, c, h, w, requires_grad=True)
        dy = torch.randn(n, c, 2 * h, 2 * w)
        mode = dict(mode="bilinear", align_corners=False) if op == "bilinear_bwd" else dict(mode="nearest")
        F.interpolate(x, scale_factor=2, **mode).backward(dy)
        ref = x.grad
        a, b = dy.permute(0, 2, 3, 1).contiguous().cuda(), torch.empty(n, h, w, c, device="cuda")
        A.check(L.srganfd_resample(2 if op == "bilinear_bwd" else 0, A.view(a), A.view(b), A.F32, n, h, w, c, st))
    else:
        x = torch.randn(n, c, h, w)
        ref = F.max_pool2d(x, 2, 2)
        a, b = x.permute(0, 2, 3, 1).contiguous().cuda(), torch.empty(n, h // 2, w // 2, c, device="cuda")
        A.check(L.srganfd_resample(3, A.view(a), A.view(b), A.F32, n, h, w, c, st))
    torch.cuda.synchronize()
    assert _rel(b.permute(0, 3, 1, 2), ref) < 1e-6


def test_adam_ema_matches_torch():
    from sr_gan_fd_amd.trainer import FlatAdamEMA
    torch.manual_seed(0)
    p0 = torch.randn(1000)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], 2e-4, (0.9, 0.999), 1e-4, 0.0)
    flat = p0.clone().cuda()
    mine = FlatAdamEMA(flat, 2e-4, (0.9, 0.999), 1e-4, 0.0, ema_decay=0.999)
    ema = None
    for it in range(3):
        g = torch.randn(1000)
        ref.grad = g.clone()
        opt.step()
        mine.step(g.cuda())
        ema = ref.detach().clone() if ema is None else (1 - 0.999) * ema + 0.999 * ref.detach()
    torch.cuda.synchronize()
    assert _rel(flat, ref) < 1e-6
    assert _rel(mine.ema, ema) < 1e-6


def test_losses_match_torch():
    from sr_gan_fd_amd import _abi as A
    torch.manual_seed(0)
    L, st = A.lib(), A.stream_ptr()
    a, b = torch.rand(2, 3, 40, 40), torch.rand(2, 3, 40, 40)
    ar = a.clone().requires_grad_(True)
    ref = 20.0 * F.l1_loss(ar, b)
    ref.backward()
    out, ws = torch.zeros(2, device="cuda"), torch.empty(A.LOSS_WS_FLOATS, device="cuda")
    ag, grad = a.cuda(), torch.empty(2, 3, 40, 40, device="cuda")
    A.check(L.srganfd_l1_loss(ag.data_ptr(), b.cuda().data_ptr(), a.numel(), 20.0, out.data_ptr(), 0, grad.data_ptr(), 20.0, None, ws.data_ptr(), st))
    torch.cuda.synchronize()
    assert abs(out[0].item() - ref.item()) < 1e-5 * abs(ref.item()) and _rel(grad, ar.grad) < 1e-6
    x = torch.randn(2, 1, 40, 40) * 3
    for target in (0.0, 1.0):
        xr = x.clone().requires_grad_(True)
        ref = 0.5 * F.binary_cross_entropy_with_logits(xr, torch.full_like(x, target))
        ref.backward()
        xg, g2 = x.cuda(), torch.empty_like(x, device="cuda")
        A.check(L.srganfd_bce_logits(xg.data_ptr(), x.numel(), target, 0.5, out.data_ptr(), 0, out.data_ptr() + 4, g2.data_ptr(), 0.5, None, ws.data_ptr(), st))
        torch.cuda.synchronize()
        assert abs(out[0].item() - ref.item()) < 1e-5 * abs(ref.item())
        assert abs(out[1].item() - torch.sigmoid(x).mean().item()) < 1e-6
        assert _rel(g2, xr.grad) < 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_discriminator(golden_dir, dtype):
    from sr_gan_fd_amd import model as M
    g = load_golden(golden_dir, "discriminator.npz")
    torch.manual_seed(0)
    d = M.discriminator_unet(in_channels=3, out_channels=1, channels=64)
    d.compute_dtype = dtype
    d.cuda().train()
    x = torch.tensor(g["x"]).cuda()
    for it in range(3):
        logits = d(x)
        e = _rel(logits, g[f"train{it}_logits"])
        print(f"D {dtype} train fwd {it}: logits err {e:.2e}")
        assert e < TOL[dtype]
        sd = d.state_dict()
        for layer in ("down_block1", "up_block1", "conv3"):   # power iteration runs in fp32 in both modes
            assert _rel(sd[f"{layer}.0.weight_u"], g[f"train{it}_{layer}_u"]) < 1e-4
            assert _rel(sd[f"{layer}.0.weight_v"], g[f"train{it}_{layer}_v"]) < 1e-4
    loss = F.binary_cross_entropy_with_logits(logits, torch.ones_like(logits))
    assert abs(loss.item() - float(g["bce_ones"])) < TOL[dtype]
    S = 65536.0 if dtype == torch.float16 else 1.0          # f16 gradients run loss-scaled, as under the reference's GradScaler
    (loss * S).backward()
    named = dict(d.named_parameters())
    worst = 0.0
    for k in ("conv1.weight", "conv4.weight", "conv4.bias", "conv3.0.weight_orig"):
        worst = max(worst, _rel(named[k].grad / S, g[f"grad/{k}"]))
    print(f"D {dtype}: worst grad err {worst:.2e}")
    assert worst < {torch.float32: 2e-3, torch.bfloat16: 1e-1, torch.float16: 2e-2}[dtype]
    if dtype == torch.float32:
        for k, want in table(g, "gsum").items():
            got = checksum(named[k].grad)
            assert np.allclose(got, want, rtol=2e-2, atol=2e-3 * abs(want[1]) + 1e-12), f"D grad checksum {k}: {got} vs {want}"
    d.eval()
    with torch.no_grad():
        assert _rel(d(x), g["eval_logits"]) < TOL[dtype]
    d.train()
    for p in d.parameters():
        p.requires_grad = False
    xin = x.clone().requires_grad_(True)
    lg = d(xin)
    assert _rel(lg, g["train3_logits"]) < TOL[dtype]
    (F.binary_cross_entropy_with_logits(lg, torch.ones_like(lg)) * S).backward()
    e = _rel(xin.grad / S, g["train3_dx"])
    e2 = _rel_l2(xin.grad / S, g["train3_dx"])
    print(f"D {dtype}: input-gradient max err {e:.2e}, L2 err {e2:.2e}")
    # bf16: every layer's gradient is re-quantised to 8 mantissa bits; bound the L2 error, report the max (f16: 11 bits)
    if dtype == torch.float32:
        assert e < 2e-3
    elif dtype == torch.float16:
        assert e2 < 3e-2 and e < 1e-1
    else:
        assert e2 < 1e-1 and e < 3e-1


def _build_gan(dtype):
    from sr_gan_fd_amd import model as M
    torch.manual_seed(0)
    d = M.discriminator_unet(in_channels=3, out_channels=1, channels=64)
    gen = M.bsrgan_x4(in_channels=3, out_channels=3, channels=64, growth_channels=32, num_rrdb=2)
    scaled_init(gen, 3.0, 0.5)
    d.compute_dtype = gen.compute_dtype = dtype
    return gen.cuda().train(), d.cuda().train()


def test_gan_steps_fused_trainer(golden_dir):
    """GanTrainer.step == two iterations of train_bsrgan.py:387-483 (golden, content loss stubbed to 0)"""
    from sr_gan_fd_amd.gan import GanTrainer
    g = load_golden(golden_dir, "gan_steps.npz")
    gen, d = _build_gan(torch.float32)
    tr = GanTrainer(gen, d, None)
    for it in range(2):
        s = tr.step(torch.tensor(g[f"it{it}_lr"]).cuda(), torch.tensor(g[f"it{it}_gt"]).cuda()).cpu().numpy()
        want = g[f"it{it}_scalars"]  # d_loss, pixel, content, adv, D(gt), D(sr)
        got = [s[0] + s[1], s[2], 0.0, s[3], s[4], s[5]]
        print(f"GAN it{it}: got {got} want {list(want)}")
        assert np.allclose(got, want, rtol=1e-3, atol=1e-5)
        assert _rel(tr.sr, g[f"it{it}_sr"]) < 1e-3
        assert _rel(gen.conv4.bias, g[f"it{it}_g_conv4_bias"]) < 1e-3
        assert _rel(d.conv4.weight, g[f"it{it}_d_probe"]) < 1e-3
        for sd, key in ((gen.state_dict(), f"it{it}_wsum_g"), (d.state_dict(), f"it{it}_wsum_d")):
            for k, want_c in table(g, key).items():
                assert np.allclose(checksum(sd[k]), want_c, rtol=2e-3, atol=2e-4 * abs(want_c[1])), f"{key} {k}"
        ema = dict(zip(tr.ge.fp.names, tr.ge.fp.grad_views(tr.g_opt.ema)))
        for k, want_c in table(g, f"it{it}_wsum_ema").items():
            if k != "n_averaged":
                assert np.allclose(checksum(ema[k[len("module."):]]), want_c, rtol=2e-3, atol=2e-4 * abs(want_c[1])), f"EMA {k}"


def test_realesrgan_generator_first_trainer(golden_dir):
    """GanTrainer(generator_first=True).step(lr, gt, gt_usm) == two iterations of Real_ESRGAN/train_realesrgan.py:407-476 run on
    Real_ESRGAN/model.py's own modules (golden; content loss stubbed to 0), from the same seeded construction."""
    from sr_gan_fd_amd import model as M
    from sr_gan_fd_amd.gan import GanTrainer
    g = load_golden(golden_dir, "realesrgan_gan_steps.npz")
    torch.manual_seed(0)
    d = M.discriminator_unet(in_channels=3, out_channels=1, channels=64)
    gen = M.rrdbnet_x4(in_channels=3, out_channels=3, channels=64, growth_channels=32, num_rrdb=2)
    scaled_init(gen, 3.0, 0.5)
    for k, want_c in table(g, "wsum_g0").items():           # Real-ESRGAN's double-draw initialisation reproduced from the seed
        assert np.allclose(checksum(gen.state_dict()[k]), want_c, rtol=1e-9, atol=1e-9 * abs(want_c[1])), k
    d.compute_dtype = gen.compute_dtype = torch.float32
    tr = GanTrainer(gen.cuda().train(), d.cuda().train(), None, g_lr=1e-4, d_lr=1e-4, betas=(0.9, 0.99), eps=1e-4, pixel_weight=1.0,
                    content_weight=[0.1, 0.1, 1.0, 1.0, 1.0], adversarial_weight=0.1, generator_first=True)
    for it in range(2):
        T = lambda k: torch.tensor(g[f"it{it}_{k}"]).cuda()
        s = tr.step(T("lr"), T("gt"), T("gt_usm")).cpu().numpy()
        want = g[f"it{it}_scalars"]  # d_loss, pixel, content, adv, sigmoid(mean D(gt)), sigmoid(mean D(sr))
        got = [s[0] + s[1], s[2], 0.0, s[3], s[4], s[5]]
        print(f"Real-ESRGAN it{it}: got {got} want {list(want)}")
        assert np.allclose(got, want, rtol=1e-3, atol=1e-5)
        assert _rel(tr.sr, g[f"it{it}_sr"]) < 1e-3
        assert _rel(gen.conv4.bias, g[f"it{it}_g_conv4_bias"]) < 1e-3
        assert _rel(d.conv4.weight, g[f"it{it}_d_probe"]) < 1e-3
        for sd, key in ((gen.state_dict(), f"it{it}_wsum_g"), (d.state_dict(), f"it{it}_wsum_d")):
            for k, want_c in table(g, key).items():
                assert np.allclose(checksum(sd[k]), want_c, rtol=2e-3, atol=2e-4 * abs(want_c[1])), f"{key} {k}"
        ema = dict(zip(tr.ge.fp.names, tr.ge.fp.grad_views(tr.g_opt.ema)))
        for k, want_c in table(g, f"it{it}_wsum_ema").items():
            if k != "n_averaged":
                assert np.allclose(checksum(ema[k[len("module."):]]), want_c, rtol=2e-3, atol=2e-4 * abs(want_c[1])), f"EMA {k}"
    with pytest.raises(Exception):
        GanTrainer(gen, d, None).step(T("lr"), T("gt"), T("gt_usm"))     # gt_usm only belongs to the generator-first order


def test_gan_steps_dropin_modules(golden_dir):
    """the reference's own loop (torch.optim.Adam, AveragedModel, autograd, retain_graph) over the drop-in modules"""
    from torch.optim.swa_utils import AveragedModel
    g = load_golden(golden_dir, "gan_steps.npz")
    gen, d = _build_gan(torch.float32)
    ema = AveragedModel(gen, avg_fn=lambda a, p, n: (1 - 0.999) * a + 0.999 * p)
    d_opt = torch.optim.Adam(d.parameters(), 2e-4, (0.9, 0.999), 1e-4, 0.0)
    g_opt = torch.optim.Adam(gen.parameters(), 8e-5, (0.9, 0.999), 1e-4, 0.0)
    bce, l1 = torch.nn.BCEWithLogitsLoss(), torch.nn.L1Loss()
    for it in range(2):
        lr, gt = torch.tensor(g[f"it{it}_lr"]).cuda(), torch.tensor(g[f"it{it}_gt"]).cuda()
        real, fake = torch.ones(2, 1, 64, 64, device="cuda"), torch.zeros(2, 1, 64, 64, device="cuda")
        for p in d.parameters():
            p.requires_grad = True
        d.zero_grad(set_to_none=True)
        gt_output = d(gt)
        d_loss_hr = bce(gt_output, real)
        d_loss_hr.backward(retain_graph=True)
        sr = gen(lr)
        sr_output = d(sr.detach().clone())
        d_loss_sr = bce(sr_output, fake)
        d_loss_sr.backward()
        d_opt.step()
        for p in d.parameters():
            p.requires_grad = False
        gen.zero_grad(set_to_none=True)
        pixel = 20.0 * l1(sr, gt)
        adv = 0.5 * bce(d(sr), real)
        (pixel + adv).backward()
        g_opt.step()
        ema.update_parameters(gen)
        got = [(d_loss_hr + d_loss_sr).item(), pixel.item(), 0.0, adv.item(), torch.sigmoid(gt_output).mean().item(),
               torch.sigmoid(sr_output).mean().item()]
        assert np.allclose(got, g[f"it{it}_scalars"], rtol=1e-3, atol=1e-5), f"{got} vs {g[f'it{it}_scalars']}"
        for sd, key in ((gen.state_dict(), f"it{it}_wsum_g"), (d.state_dict(), f"it{it}_wsum_d"), (ema.state_dict(), f"it{it}_wsum_ema")):
            for k, want_c in table(g, key).items():
                if k != "n_averaged":
                    assert np.allclose(checksum(sd[k]), want_c, rtol=2e-3, atol=2e-4 * abs(want_c[1])), f"{key} {k}"


def test_gan_steps_f16_trainer_vs_reference(golden_dir):
    """The benchmark dtype against the reference's own two GAN iterations (gan_steps.npz, the vectors the f32 test above meets at
    1e-3): f16 activations / weights / loss-scaled gradients, fp32 master weights.  Asserted bounds: every logged scalar (d_loss, pixel,
    adversarial, D(gt), D(sr)) within 1e-3 relative (north_star's tolerance; one f16 rounding is 4.9e-4, the logits of the 14-layer
    U-Net carry a few of them and still meet it); SR pixels within 1e-3 of the range; probed parameters after the Adam steps within 5e-3 of each tensor's update
    size-independent checksum tolerance (Adam's g / (sqrt(v) + eps) amplifies relative gradient error where |g| ~ eps = 1e-4)."""
    from sr_gan_fd_amd.gan import GanTrainer
    g = load_golden(golden_dir, "gan_steps.npz")
    gen, d = _build_gan(torch.float16)
    tr = GanTrainer(gen, d, None)
    assert tr.scaler.enabled and tr.scaler.scale == 65536.0
    for it in range(2):
        s = tr.step(torch.tensor(g[f"it{it}_lr"]).cuda(), torch.tensor(g[f"it{it}_gt"]).cuda()).cpu().numpy()
        want = g[f"it{it}_scalars"]  # d_loss, pixel, content, adv, D(gt), D(sr)
        got = [s[0] + s[1], s[2], 0.0, s[3], s[4], s[5]]
        err = max(abs(a - b) / max(abs(b), 1e-6) for a, b in zip(got, want) if b != 0.0)
        print(f"f16 GAN it{it}: got {got} want {list(want)} worst rel {err:.2e}, SR err {_rel(tr.sr, g[f'it{it}_sr']):.2e}")
        assert err < 1e-3
        assert _rel(tr.sr, g[f"it{it}_sr"]) < 1e-3
        print(f"   parameter probes: G conv4.bias {_rel(gen.conv4.bias, g[f'it{it}_g_conv4_bias']):.2e}, D conv4.weight {_rel(d.conv4.weight, g[f'it{it}_d_probe']):.2e}")
        assert _rel(gen.conv4.bias, g[f"it{it}_g_conv4_bias"]) < 5e-3
        assert _rel(d.conv4.weight, g[f"it{it}_d_probe"]) < 5e-3
    rep = tr.scaler.report()
    assert rep["optimizer_steps"] == 4 and rep["skipped"] == 0, rep


def test_gan_iteration_at_the_bsrgan_default_crop_f16_vs_oracle():
    """bsrgan_config.py:62,101-102: x2, LR crops of 72 x 72 (GT 144 x 144), the 23-RRDB generator -- the ragged case of every tile shape of the
    engines (72 = 4.5 x 16 rows = 2.25 x 32 columns in the trunk, 144 = 4.5 x 32 in the tail, the discriminator and its stride-2 stages at
    144 / 72 / 36 / 18) and the shape whose dense blocks stay on the per-layer launches.  One GAN iteration (train_bsrgan.py:387-483,
    content term off: no VGG weights in the reference tree) in float16, the scripts' autocast dtype, against the fp32 CPU oracle; batch 4
    instead of 16 keeps the oracle at seconds (tiles are per image: the launches differ in count only).  Asserted: every logged scalar
    within 1e-3 relative, probed parameters after both Adam steps within 5e-3 of the tensor's range (as the fixture-based f16 test)."""
    from oracle import srgan_oracle as O
    from sr_gan_fd_amd import model as M
    from sr_gan_fd_amd.gan import GanTrainer
    torch.manual_seed(9)
    lr_img, gt = torch.rand(4, 3, 72, 72), torch.rand(4, 3, 144, 144)
    torch.manual_seed(0)
    d = M.discriminator_unet(in_channels=3, out_channels=1, channels=64)
    gen = M.bsrgan_x2(in_channels=3, out_channels=3, channels=64, growth_channels=32, num_rrdb=23)
    scaled_init(gen, 3.0, 0.5)
    G, D = sd_to_params(gen.state_dict()), sd_to_params(d.state_dict(), d=True)
    g_opt, d_opt = O.AdamState(G, O.g_param_names(G)), O.AdamState(D, O.d_param_names(D))
    out = O.gan_step(G, D, g_opt, d_opt, lr_img, gt, upscale=2, g_lr=8e-5, d_lr=2e-4, betas=(0.9, 0.999), eps=1e-4, pixel_weight=20.0,
                     content_weight=1.0, adversarial_weight=0.5)
    d.compute_dtype = gen.compute_dtype = torch.float16
    gen, d = gen.cuda().train(), d.cuda().train()
    tr = GanTrainer(gen, d, None)
    s = tr.step(lr_img.cuda(), gt.cuda()).cpu().numpy()
    got = [s[0] + s[1], s[2], s[3], s[4], s[5]]
    want = [out[k] for k in ("d_loss", "pixel_loss", "adversarial_loss", "d_gt_probability", "d_sr_probability")]
    err = max(abs(a - b) / max(abs(b), 1e-6) for a, b in zip(got, want))
    e_g, e_d = _rel(gen.conv4.bias, G["conv4.bias"]), _rel(d.conv4.weight, D["conv4.weight"])
    print(f"f16 GAN iteration at 72 -> 144 x2, 23 RRDB: got {got} want {want} worst rel {err:.2e}; G conv4.bias {e_g:.2e}, D conv4.weight {e_d:.2e}")
    assert err < 1e-3 and e_g < 5e-3 and e_d < 5e-3
    rep = tr.scaler.report()
    assert rep["optimizer_steps"] == 2 and rep["skipped"] == 0, rep


def test_loss_scaler_skips_nonfinite_step():
    """GradScaler semantics (train_bsrgan.py:436-437,466-467): a non-finite gradient leaves parameters and Adam state untouched,
    halves the scale, and the EMA still advances; finite steps update as usual and leave the scale alone."""
    from sr_gan_fd_amd.trainer import FlatAdamEMA, LossScaler
    torch.manual_seed(0)
    p = torch.randn(1000, device="cuda")
    opt = FlatAdamEMA(p, 1e-2, (0.9, 0.999), 1e-8, 0.0, ema_decay=0.999)
    sc = LossScaler(p.device, enabled=True, init_scale=1024.0)
    g = torch.randn(1000, device="cuda")
    p0 = p.clone()
    sc.step(opt, g * 1024.0, 1.0)                     # finite: an ordinary Adam step on g (the kernel unscales with the device's 1 / scale)
    ref = torch.nn.Parameter(p0.clone())
    o = torch.optim.Adam([ref], 1e-2, (0.9, 0.999), 1e-8)
    ref.grad = g.clone()
    o.step()
    assert torch.allclose(p, ref.data, rtol=1e-5, atol=1e-7)
    p1, m1, ema1 = p.clone(), opt.m.clone(), opt.ema.clone()
    bad = g.clone()
    bad[123] = float("inf")
    sc.step(opt, bad, 1.0)                            # non-finite: skipped on the device
    torch.cuda.synchronize()
    assert torch.equal(p, p1) and torch.equal(opt.m, m1) and int(opt.step_dev.item()) == 1
    assert torch.allclose(opt.ema, 0.001 * ema1 + 0.999 * p1)
    rep = sc.report()
    assert rep == {"enabled": True, "scale": 512.0, "optimizer_steps": 2, "skipped": 1}, rep
    nan = g.clone()
    nan[7] = float("nan")
    sc.step(opt, nan, 1.0)
    assert sc.report()["scale"] == 256.0 and torch.equal(p, p1)
    # the very next step already runs at the backed-off scale (no lag): a gradient scaled by 256 is unscaled by 1 / 256
    sc.step(opt, g * 256.0, 1.0)
    ref.grad = g.clone()
    o.step()
    assert torch.allclose(p, ref.data, rtol=1e-5, atol=1e-7)


def test_loss_scaler_follows_torch_gradscaler():
    """A forced-overflow / growth sequence through torch.amp.GradScaler (the reference's object, train_bsrgan.py:109,436-437,466-467)
    and through LossScaler: the same steps are skipped and the scale is the same after EVERY update -- overflows at steps 3, 4 and 9
    (two in a row: two backoffs, one skipped step each), growth after every 5 clean steps (growth_interval shortened from 2000),
    and a growth right after a backoff's tracker reset."""
    from sr_gan_fd_amd.trainer import FlatAdamEMA, LossScaler
    torch.manual_seed(3)
    dev = torch.device("cuda")
    kw = dict(init_scale=1024.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=5)
    ts = torch.amp.GradScaler("cuda", **kw)
    ts.scale(torch.ones(1, device=dev))          # torch creates its device-side scale lazily, at the first scale(loss)
    tp = torch.nn.Parameter(torch.randn(256, device=dev))
    topt = torch.optim.Adam([tp], 1e-2, (0.9, 0.999), 1e-8)
    p = tp.detach().clone()
    opt = FlatAdamEMA(p, 1e-2, (0.9, 0.999), 1e-8, 0.0, ema_decay=None)
    sc = LossScaler(dev, enabled=True, **kw)
    bad_steps = {3, 4, 9, 27}
    skipped_t, skipped_s = [], []
    for k in range(40):
        g = torch.randn(256, device=dev)
        S_t, S_s = ts.get_scale(), sc.scale
        assert S_t == S_s, (k, S_t, S_s)                       # the scale the backward pass of step k is seeded with
        gt_, gs_ = g * S_t, g * S_s
        if k in bad_steps:
            gt_[5] = float("inf"); gs_[5] = float("inf")
        before_t, before_s = tp.detach().clone(), p.clone()
        tp.grad = gt_
        ts.step(topt)                                          # unscale_ + found_inf + (skipped) step
        ts.update()
        sc.step(opt, gs_, 1.0, update_ema=False)
        torch.cuda.synchronize()
        skipped_t.append(bool(torch.equal(tp.detach(), before_t)))
        skipped_s.append(bool(torch.equal(p, before_s)))
        assert torch.allclose(p, tp.detach(), rtol=1e-5, atol=1e-7), k
    assert skipped_t == skipped_s and [k for k, v in enumerate(skipped_s) if v] == sorted(bad_steps)
    assert ts.get_scale() == sc.scale
    rep = sc.report()
    assert rep["optimizer_steps"] == 40 and rep["skipped"] == len(bad_steps)
    assert sc.state_dict()["_growth_tracker"] == ts.state_dict()["_growth_tracker"]


def test_trainer_refuses_f16_set_after_construction():
    """compute_dtype is a plain attribute: flipped to float16 after the trainer was built, the disabled scaler would let the gradients
    underflow silently -- the step raises instead"""
    from sr_gan_fd_amd import _abi as A
    from sr_gan_fd_amd.gan import GanTrainer
    gen, d = _build_gan(torch.float32)
    tr = GanTrainer(gen, d, None)
    gen.compute_dtype = d.compute_dtype = torch.float16
    with pytest.raises(A.SrganfdError, match="loss scaler is disabled"):
        tr.step(torch.rand(2, 3, 16, 16).cuda(), torch.rand(2, 3, 64, 64).cuda())


def test_gan_step_bf16_runs_close():
    """bf16 (benchmark dtype) GAN iteration stays within 5e-2 of the f32 iteration on the same inputs"""
    from sr_gan_fd_amd.gan import GanTrainer
    torch.manual_seed(5)
    lr, gt = torch.rand(2, 3, 16, 16).cuda(), torch.rand(2, 3, 64, 64).cuda()
    outs = {}
    for dt in (torch.float32, torch.bfloat16):
        gen, d = _build_gan(dt)
        outs[dt] = GanTrainer(gen, d, None).step(lr, gt).cpu().numpy()[:6].copy()
    print("GAN scalars f32 ", outs[torch.float32], "\nGAN scalars bf16", outs[torch.bfloat16])
    assert np.allclose(outs[torch.bfloat16], outs[torch.float32], rtol=5e-2, atol=5e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_content_loss_vs_oracle(dtype):
    """VGG-19 taps (parity pinned only against the CPU oracle: no torchvision / ImageNet weights offline; tools/pin_vgg.py writes the
    fixture test_content_loss_vs_pinned_torchvision_taps uses where both exist).  Asserted bounds on the (1, 5) value, max error
    over the scale: f32 1e-3; f16 -- the benchmarked dtype -- 2e-3 against the oracle run on the same f16-rounded conv weights (what is
    left is the f16 rounding of 16 layers of stored activations); bf16 5e-2 against the fp32 oracle."""
    from oracle import srgan_oracle as O
    from sr_gan_fd_amd import model as M
    nodes = ["features.2", "features.7", "features.16", "features.25", "features.34"]
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    cl = M.ContentLoss(nodes, mean, std)
    cl.compute_dtype = dtype
    torch.manual_seed(3)
    sr, gt = torch.rand(2, 3, 32, 48), torch.rand(2, 3, 32, 48)
    P = {"features." + k: v.detach().clone() for k, v in cl.features.state_dict().items()}
    want = O.content_loss(sr, gt, rounded_weights(P, dtype) if dtype == torch.float16 else P, nodes, mean, std, taps_post_relu=True)
    cl.cuda()
    got = cl(sr.cuda(), gt.cuda())
    assert got.shape == (1, 5) and not got.requires_grad
    e = _rel(got, want)
    e_each = ((got.cpu() - want).abs() / want.abs()).max().item()
    print(f"content loss {dtype}: {got.cpu().numpy()} vs {want.numpy()} err {e:.2e} (worst node, relative to its own value: {e_each:.2e})")
    assert e < {torch.float32: 1e-3, torch.float16: 2e-3, torch.bfloat16: 5e-2}[dtype]
    if dtype == torch.float16:
        assert e_each < 2e-3            # north_star's 1e-3 on loss values is met per node with margin 2 (observed: see the printed line)
    if dtype == torch.float32:
        cl2 = M.ContentLoss(nodes, mean, std, taps_post_relu=False)
        cl2.compute_dtype = dtype
        cl2.cuda()
        want2 = O.content_loss(sr, gt, P, nodes, mean, std, taps_post_relu=False)
        assert _rel(cl2(sr.cuda(), gt.cuda()), want2) < 1e-3


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_content_loss_vs_pinned_torchvision_taps(golden_dir, dtype):
    """The HIP ContentLoss on the ImageNet VGG-19 weights against values produced by the reference's own class (fixture of
    tools/pin_vgg.py).  Runs only where that fixture and the weights exist (tests.util.pinned_vgg; skipped on the build's GPU box: row
    A7 stays "parity unpinned" until someone with torchvision runs the script).  Bounds: 1e-3 (f32), 2e-3 (f16) per node."""
    from sr_gan_fd_amd import model as M
    g, _, wpath = pinned_vgg(golden_dir)
    nodes, mean, std = [str(n) for n in g["nodes"]], [float(v) for v in g["mean"]], [float(v) for v in g["std"]]
    cl = M.ContentLoss(nodes, mean, std, weights_path=wpath)
    cl.compute_dtype = dtype
    cl.cuda()
    got = cl(torch.tensor(g["sr"]).cuda(), torch.tensor(g["gt"]).cuda()).cpu().numpy()
    err = np.abs(got - g["bsrgan_values"]) / np.abs(g["bsrgan_values"])
    print(f"pinned content loss {dtype}: {got} vs {g['bsrgan_values']} rel err {err}")
    assert err.max() < (1e-3 if dtype == torch.float32 else 2e-3)
    cl1 = M.ContentLoss("features.34", mean, std, weights_path=wpath)
    cl1.compute_dtype = dtype
    cl1.cuda()
    s = torch.tensor(g["sr"]).cuda().requires_grad_(True)
    v = cl1(s, torch.tensor(g["gt"]).cuda())
    v.backward()
    assert abs(v.item() - float(g["esrgan_value"])) < (1e-4 if dtype == torch.float32 else 2e-3) * abs(float(g["esrgan_value"]))
    ref = torch.tensor(g["esrgan_dsr"]).double()
    assert ((s.grad.double().cpu() - ref).norm() / ref.norm()).item() < (1e-2 if dtype == torch.float32 else 2e-1)


def test_content_loss_floor_pooling_at_reference_crop_size():
    """aesrgan_config trains on 120x120 crops: 120 -> 60 -> 30 -> 15 -> 7 through the four max-pools (floor, like torch)"""
    from oracle import srgan_oracle as O
    from sr_gan_fd_amd import model as M
    nodes = ["features.2", "features.7", "features.16", "features.25", "features.34"]
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    cl = M.ContentLoss(nodes, mean, std)
    cl.compute_dtype = torch.float32
    torch.manual_seed(5)
    sr, gt = torch.rand(1, 3, 120, 104), torch.rand(1, 3, 120, 104)
    P = {"features." + k: v.detach().clone() for k, v in cl.features.state_dict().items()}
    want = O.content_loss(sr, gt, P, nodes, mean, std, taps_post_relu=True)
    cl.cuda()
    assert _rel(cl(sr.cuda(), gt.cuda()), want) < 1e-3


def test_data_parallel_shards_equal_full_batch():
    """N-rank data-parallel step == single-rank step on the concatenated batch.  Checked on
    one GPU: the mean of the two shard gradients (what all-reduce(sum) x 1/2 produces) equals the
    full-batch gradient of the generator (f32 mode)."""
    from sr_gan_fd_amd import model as M
    from sr_gan_fd_amd.engine import generator_engine
    torch.manual_seed(0)
    gen = M.bsrgan_x4(num_rrdb=1)
    scaled_init(gen, 3.0, 0.5)
    gen.compute_dtype = torch.float32
    gen.cuda()
    lr, gt = torch.rand(4, 3, 16, 16).cuda(), torch.rand(4, 3, 64, 64).cuda()

    def flat_grad(x, y):
        gen.zero_grad(set_to_none=True)
        F.l1_loss(gen(x), y).backward()
        return torch.cat([p.grad.reshape(-1) for p in gen.parameters()]).clone()
    full = flat_grad(lr, gt)
    halves = 0.5 * (flat_grad(lr[:2], gt[:2]) + flat_grad(lr[2:], gt[2:]))
    assert _rel(halves, full) < 1e-4


def test_train_generator_false_freezes_generator_only():
    """bsrgan_config.train_generator = False (train_bsrgan.py:460): every forward still runs, D steps, G and its EMA do not"""
    from sr_gan_fd_amd.gan import GanTrainer
    gen, d = _build_gan(torch.float32)
    tr = GanTrainer(gen, d, None, train_generator=False)
    g0, d0 = tr.g_opt.flat.clone(), tr.d_opt.flat.clone()
    s = tr.step(torch.rand(2, 3, 16, 16, device="cuda"), torch.rand(2, 3, 64, 64, device="cuda")).cpu().numpy()
    assert np.isfinite(s).all() and s[2] > 0 and s[3] > 0          # pixel and adversarial losses were evaluated
    assert torch.equal(tr.g_opt.flat, g0) and not torch.equal(tr.d_opt.flat, d0)
    assert tr.g_opt.t == 0 and tr.d_opt.t == 1


def test_stale_activations_are_refused_not_silently_used():
    """a second training forward of the same network overwrites the saved activations: backward through the first one
    must fail loudly instead of producing a gradient from the wrong activations"""
    from sr_gan_fd_amd import _abi as A
    gen, _ = _build_gan(torch.float32)
    x1 = torch.rand(1, 3, 16, 16, device="cuda")
    y1 = gen(x1)
    y2 = gen(torch.rand(1, 3, 16, 16, device="cuda"))
    with pytest.raises(A.SrganfdError):
        y1.sum().backward()
    y2.sum().backward()                                            # the latest forward is still valid
    assert all(p.grad is not None for p in gen.parameters())


def test_two_rank_gan_iteration_equals_full_batch_iteration():
    """SURVEY 8(e) for the whole GAN iteration: two data-parallel ranks (two trainers in two threads of this process, their
    all-reduce emulated by a barrier + sum) on two 2-image shards == one trainer on the 4-image batch (f32 mode)."""
    import threading
    from sr_gan_fd_amd.gan import GanTrainer

    class Exchange:
        def __init__(self, world):
            self.world, self.slots, self.bar = world, {}, threading.Barrier(world)

        def allreduce(self, rank, t):
            self.slots[rank] = t
            self.bar.wait()
            torch.cuda.synchronize()
            total = sum(self.slots[r] for r in range(self.world))
            self.bar.wait()
            t.copy_(total)
            self.bar.wait()
            return 1.0 / self.world

    torch.manual_seed(3)
    lr, gt = torch.rand(4, 3, 16, 16).cuda(), torch.rand(4, 3, 64, 64).cuda()
    gen, d = _build_gan(torch.float32)
    ref = GanTrainer(gen, d, None)
    s_ref = ref.step(lr, gt).cpu().numpy().copy()
    ex = Exchange(2)
    trainers, scalars, errors = [], [None, None], []
    for r in range(2):
        g_r, d_r = _build_gan(torch.float32)
        t = GanTrainer(g_r, d_r, None)
        t._allreduce = (lambda grad, r=r: ex.allreduce(r, grad))          # the discriminator's exchange

        class Buckets:                                                      # the generator's: buckets during backward, sum at finish()
            def __init__(self, r):
                self.r, self.flat, self.covered = r, None, 0

            def begin(self):
                self.covered = 0

            def bucket(self, flat, lo, hi):
                self.flat, self.covered = flat, self.covered + hi - lo

            def finish(self):
                assert self.covered == self.flat.numel()
                return ex.allreduce(self.r, self.flat)
        t.g_reducer = Buckets(r)
        trainers.append(t)

    def run(r):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                scalars[r] = trainers[r].step(lr[2 * r:2 * r + 2].contiguous(), gt[2 * r:2 * r + 2].contiguous()).cpu().numpy().copy()
        except Exception as e:      # noqa: BLE001
            errors.append(e)
            ex.bar.abort()
    th = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not errors, errors
    torch.cuda.synchronize()
    # every rank ends with the same parameters as the single-process step on the concatenated batch
    for t in trainers:
        assert _rel(t.g_opt.flat, ref.g_opt.flat) < 1e-4 and _rel(t.d_opt.flat, ref.d_opt.flat) < 1e-4
    assert torch.equal(trainers[0].g_opt.flat, trainers[1].g_opt.flat) and torch.equal(trainers[0].d_opt.flat, trainers[1].d_opt.flat)
    # the logged scalars are shard means: their average is the full-batch value
    assert np.allclose(0.5 * (scalars[0] + scalars[1])[:6], s_ref[:6], rtol=1e-4, atol=1e-6)


def test_fused_gan_trainer_discriminator_learns():
    """over a fixed batch the discriminator must pull D(gt) and D(sr) apart (bf16, the benchmark dtype): sign / scale errors in the
    BCE gradient, the spectral-norm backward or the D Adam step would stall this"""
    from sr_gan_fd_amd.gan import GanTrainer
    gen, d = _build_gan(torch.bfloat16)
    tr = GanTrainer(gen, d, None, d_lr=1e-3)
    torch.manual_seed(9)
    lr = torch.rand(2, 3, 16, 16, device="cuda")
    gt = F.interpolate(lr, scale_factor=4, mode="bicubic", align_corners=False).clamp(0, 1)
    hist = np.stack([tr.step(lr, gt).cpu().numpy().copy() for _ in range(25)])
    d_loss = hist[:, 0] + hist[:, 1]
    print("d_loss", np.round(d_loss[::6], 4), "D(gt)", np.round(hist[::6, 4], 3), "D(sr)", np.round(hist[::6, 5], 3))
    assert np.isfinite(hist).all()
    assert d_loss[-3:].mean() < d_loss[0] - 0.1 and d_loss[-1] < d_loss[12] < d_loss[0]      # steady descent from ~2 ln 2
    assert hist[-1, 4] > hist[-1, 5]                      # real scored above fake


def test_side_stream_discriminator_update_equals_inline():
    """Data-parallel runs put D's all-reduce + Adam on a side stream beside the pixel loss / VGG forwards (parallel.SideStreamReducer);
    the iteration must not change by a bit: same two GAN iterations with the update inline and on the side stream (the collective
    itself replaced by the identity: one rank)."""
    from sr_gan_fd_amd.gan import GanTrainer
    from sr_gan_fd_amd.parallel import SideStreamReducer
    torch.manual_seed(11)
    batches = [(torch.rand(2, 3, 16, 16).cuda(), torch.rand(2, 3, 64, 64).cuda()) for _ in range(2)]
    outs = []
    for side in (False, True):
        gen, d = _build_gan(torch.float32)
        tr = GanTrainer(gen, d, None)
        if side:
            tr.d_reducer = SideStreamReducer(torch.device("cuda"), pg="one-rank stand-in")
            assert tr.d_reducer.stream is not None
        sc = [tr.step(x, y).cpu().numpy().copy() for x, y in batches]
        torch.cuda.synchronize()
        outs.append((sc, tr.g_opt.flat.clone(), tr.d_opt.flat.clone(), tr.sr.clone()))
    for a, b in zip(outs[0][0], outs[1][0]):
        assert np.array_equal(a, b)
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2]) and torch.equal(outs[0][3], outs[1][3])


def test_bucketed_generator_exchange_equals_inline(monkeypatch):
    """Data-parallel runs send the generator's flat gradient in three buckets on a side stream while its backward pass is still
    running (parallel.BucketReducer; the engine marks contiguous ranges final: tail, upper half of the trunk, the rest).  With the
    collective replaced by a kernel that rewrites the range in place on that stream (one rank: the sum is the value), two
    generator-only iterations and two GAN iterations must not change by a bit, and the buckets must cover the buffer exactly once."""
    from sr_gan_fd_amd import model as M, parallel as P
    from sr_gan_fd_amd.gan import GanTrainer
    from sr_gan_fd_amd.trainer import GeneratorTrainer
    seen = []

    def fake_all_reduce(t, op=None, group=None):
        seen.append((t.data_ptr(), t.numel(), torch.cuda.current_stream().cuda_stream))
        t.mul_(1.0)                                   # a kernel on the reducer's stream that reads and rewrites the whole range
    monkeypatch.setattr(P.dist, "all_reduce", fake_all_reduce)
    monkeypatch.setattr(P.dist, "get_world_size", lambda pg=None: 1)
    # BucketReducer asks the group who shares its GPU (parallel.dense_chain_needs_its_own_gpu): the one-rank stand-in answers with itself
    monkeypatch.setattr(P.dist, "all_gather_object", lambda out, obj, group=None: out.__setitem__(slice(None), [obj]))
    torch.manual_seed(12)
    batches = [(torch.rand(2, 3, 16, 16).cuda(), torch.rand(2, 3, 64, 64).cuda()) for _ in range(2)]

    def gen5():
        torch.manual_seed(0)
        g = M.bsrgan_x4(in_channels=3, out_channels=3, channels=64, growth_channels=32, num_rrdb=2)     # 6 dense blocks: all three buckets
        scaled_init(g, 3.0, 0.5)
        g.compute_dtype = torch.float32
        return g.cuda().train()
    outs = []
    for bucketed in (False, True):
        tr = GeneratorTrainer(gen5(), lr=1e-4)
        if bucketed:
            tr.g_reducer = P.BucketReducer(torch.device("cuda"), pg="one-rank stand-in")
            assert tr.g_reducer.stream is not None
        losses = [tr.step(x, y).item() for x, y in batches]
        torch.cuda.synchronize()
        outs.append((losses, tr.flat.clone(), tr.opt.ema.clone()))
        if bucketed:
            main = torch.cuda.current_stream().cuda_stream
            per_step = seen[:len(seen) // 2]
            assert len(per_step) == 3 and sum(n for _, n, _ in per_step) == tr.flat.numel() and tr.g_reducer.sizes == [n for _, n, _ in per_step]
            assert all(s != main for _, _, s in seen)
            # disjoint, contiguous, tail first
            spans = sorted((p, p + 4 * n) for p, n, _ in per_step)
            assert spans[0][1] == spans[1][0] and spans[1][1] == spans[2][0] and per_step[0][0] == spans[2][0]
    assert outs[0][0] == outs[1][0] and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    seen.clear()
    gouts = []
    for bucketed in (False, True):
        gen, d = _build_gan(torch.float32)
        tr = GanTrainer(gen, d, None)
        if bucketed:
            tr.g_reducer = P.BucketReducer(torch.device("cuda"), pg="one-rank stand-in")
        sc = [tr.step(x, y).cpu().numpy().copy() for x, y in batches]
        torch.cuda.synchronize()
        gouts.append((sc, tr.g_opt.flat.clone(), tr.d_opt.flat.clone()))
    assert all(np.array_equal(a, b) for a, b in zip(gouts[0][0], gouts[1][0]))
    assert torch.equal(gouts[0][1], gouts[1][1]) and torch.equal(gouts[0][2], gouts[1][2]) and len(seen) > 0


def test_f16_training_run_tracks_the_f32_run():
    """Forty GAN iterations in f16 (the benchmarked dtype: loss-scaled gradients, skip-on-overflow Adam, fp32 master weights) next to
    the same forty in f32 on the same batches: no skipped step, every scalar finite, the loss scale untouched, and the two trajectories
    stay together (pixel loss within 1e-3 over the first ten iterations and within 5 % over all forty -- Adam at lr 2e-4 turns one flipped
    L1 sign into a visible difference after a few dozen steps -- generator weights within 15 % of the distance the run moved them)."""
    from sr_gan_fd_amd.gan import GanTrainer
    gen_b = torch.Generator(device="cuda").manual_seed(21)
    base = torch.rand(4, 3, 8, 8, device="cuda", generator=gen_b)
    gts = [F.interpolate(base + 0.05 * torch.rand(4, 3, 8, 8, device="cuda", generator=gen_b), size=(64, 64), mode="bilinear").clamp(0, 1) for _ in range(4)]
    lrs = [F.interpolate(g, size=(16, 16), mode="bilinear") for g in gts]
    runs = {}
    for dt in (torch.float32, torch.float16):
        gen, d = _build_gan(dt)
        tr = GanTrainer(gen, d, None, g_lr=2e-4, d_lr=2e-4)
        w0 = tr.g_opt.flat.clone()
        hist = []
        for it in range(40):
            hist.append(tr.step(lrs[it % 4], gts[it % 4]).cpu().numpy()[:6].copy())
        torch.cuda.synchronize()
        runs[dt] = (np.stack(hist), tr.g_opt.flat.clone(), w0, tr.scaler.report())
    h32, w32, w0, _ = runs[torch.float32]
    h16, w16, _, rep = runs[torch.float16]
    assert np.isfinite(h16).all() and np.isfinite(h32).all()
    assert rep == {"enabled": True, "scale": 65536.0, "optimizer_steps": 80, "skipped": 0}, rep
    assert h32[-4:, 2].mean() < 0.8 * h32[:4, 2].mean()                       # the pixel loss (index 2) went down: it is a training run
    assert np.allclose(h16[:10, 2], h32[:10, 2], rtol=1e-3), np.abs(h16[:10, 2] / h32[:10, 2] - 1).max()
    assert np.allclose(h16[:, 2], h32[:, 2], rtol=5e-2), np.abs(h16[:, 2] / h32[:, 2] - 1).max()
    moved = (w32 - w0).norm().item()
    drift = (w16 - w32).norm().item()
    print(f"f16 vs f32 after 40 GAN iterations: pixel loss {h16[-1, 2]:.5f} vs {h32[-1, 2]:.5f}, weight drift {drift:.3e} of {moved:.3e} moved")
    assert drift < 0.15 * moved
