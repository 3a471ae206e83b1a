#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING the reference (build container only).

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

This is the only file in the repository that reads /root/reference.  The reference needs
``torchvision`` (absent here) only for ContentLoss, so empty stub modules are injected before the
import; nothing of the reference is copied -- only inputs/outputs (data) are saved.
Weights are NOT stored (too large): every fixture is built under ``torch.manual_seed(seed)`` with
the reference's own constructors, and the tests rebuild identical weights from the same seed with
this repository's module containers; per-tensor checksums stored here prove the rebuild is exact.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("SRGAN_REFERENCE", "/root/reference")


def _stub_torchvision():
    for name in ("torchvision", "torchvision.models", "torchvision.transforms",
                 "torchvision.models.feature_extraction"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision.models.feature_extraction"].create_feature_extractor = None


def _stub_basicsr():
    for name in ("basicsr", "basicsr.utils", "basicsr.utils.registry"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)

    class _Reg:
        def register(self, *a, **k):
            return (lambda c: c) if not a else a[0]
    sys.modules["basicsr.utils.registry"].ARCH_REGISTRY = _Reg()


def load_ref(subdir):
    _stub_torchvision()
    _stub_basicsr()
    path = os.path.join(REF, subdir)
    sys.path.insert(0, path)
    try:
        sys.modules.pop("model", None)
        sys.modules.pop("aesrgan_config", None)
        mod = importlib.import_module("model")
    finally:
        sys.path.remove(path)
        sys.modules.pop("model", None)
        sys.modules.pop("aesrgan_config", None)
    return mod


def checksum(t):
    """(sum, abs-sum, position-weighted sum) in float64 -- catches value AND ordering errors."""
    a = t.detach().double().flatten()
    w = torch.cos(torch.arange(a.numel(), dtype=torch.float64) * 0.37)
    return np.array([a.sum().item(), a.abs().sum().item(), (a * w).sum().item()])


def sd_checksums(sd):
    return {k: checksum(v) for k, v in sd.items() if torch.is_tensor(v) and v.dtype.is_floating_point}


def scaled_init(g, scale, bias):
    """SURVEY 8c init recipe: default init is degenerate (SR std 5e-5, 35 % clamped)."""
    with torch.no_grad():
        for p in g.parameters():
            if p.dim() == 4:
                p.mul_(scale)
        g.conv4.bias.fill_(bias)


def np_(t):
    return t.detach().cpu().numpy().copy()  # copy: buffers are later updated in place


def save(name, **arrs):
    flat = {}
    for k, v in arrs.items():
        if isinstance(v, dict):  # checksum tables: one (n,3) array + the key list
            flat[k] = np.stack([np.asarray(vv) for vv in v.values()]) if v else np.zeros((0, 3))
            flat[k + "__keys"] = np.array(list(v.keys()))
        else:
            flat[k] = np.asarray(v)
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **flat)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} kB, {len(flat)} arrays")


# ------------------------------------------------------------------------------------------
def gold_blocks(M):
    out = {}
    for kind, ctor in (("rdb", M._ResidualDenseBlock), ("rrdb", M._ResidualResidualDenseBlock)):
        torch.manual_seed(0)
        blk = ctor(64, 32)
        x = torch.randn(1, 64, 16, 16, requires_grad=True)
        r = torch.randn(1, 64, 16, 16)
        y = blk(x)
        (y * r).sum().backward()
        out[f"{kind}_x"] = np_(x)
        out[f"{kind}_r"] = np_(r)
        out[f"{kind}_y"] = np_(y)
        out[f"{kind}_dx"] = np_(x.grad)
        out[f"{kind}_wsum"] = sd_checksums(blk.state_dict())
        out[f"{kind}_gsum"] = {k: checksum(p.grad) for k, p in blk.named_parameters()}
        first = "conv1" if kind == "rdb" else "rdb1.conv1"
        last = "conv5" if kind == "rdb" else "rdb3.conv5"
        named = dict(blk.named_parameters())
        out[f"{kind}_g_first_w"] = np_(named[first + ".weight"].grad)
        out[f"{kind}_g_last_b"] = np_(named[last + ".bias"].grad)
    save("blocks.npz", **out)


def gold_generator(MB, ME):
    out = {}
    cases = [
        ("bsrgan_x4_r2_s3", MB.bsrgan_x4, dict(num_rrdb=2), 4, 3.0, (2, 3, 16, 16)),
        ("bsrgan_x2_r2_s3", MB.bsrgan_x2, dict(num_rrdb=2), 2, 3.0, (2, 3, 16, 16)),
        ("bsrgan_x4_r2_s5", MB.bsrgan_x4, dict(num_rrdb=2), 4, 5.0, (2, 3, 16, 16)),
        ("rrdbnet_x4_r23_s3", ME.rrdbnet_x4, dict(num_blocks=23), 4, 3.0, (1, 3, 16, 16)),
        ("bsrgan_x4_r23_s3_odd", MB.bsrgan_x4, dict(num_rrdb=23), 4, 3.0, (1, 3, 12, 20)),
    ]
    for name, fac, kw, s, scale, shape in cases:
        torch.manual_seed(0)
        g = fac(in_channels=3, out_channels=3, channels=64, growth_channels=32, **kw)
        scaled_init(g, scale, 0.5)
        x = torch.rand(*shape)
        gt = torch.rand(shape[0], 3, shape[2] * s, shape[3] * s)
        sr = g(x)
        loss = torch.nn.functional.l1_loss(sr, gt)
        loss.backward()
        out[f"{name}/x"] = np_(x)
        out[f"{name}/gt"] = np_(gt)
        out[f"{name}/sr"] = np_(sr)
        out[f"{name}/loss"] = np.array(loss.item())
        out[f"{name}/clamped01"] = np.array([(sr == 0).float().mean().item(), (sr == 1).float().mean().item()])
        named = dict(g.named_parameters())
        out[f"{name}/wsum"] = sd_checksums(g.state_dict())
        out[f"{name}/gsum"] = {k: checksum(p.grad) for k, p in named.items()}
        for k in ("conv1.weight", "conv4.weight", "conv4.bias", "trunk.0.rdb1.conv1.bias",
                  "trunk.1.rdb3.conv5.bias", "conv2.bias"):
            out[f"{name}/grad/{k}"] = np_(named[k].grad)
    save("generator.npz", **out)


def gold_realesrgan_rrdbnet(MR):
    """Real_ESRGAN/model.py:179-262 at the factors its class supports beyond the shipped x4 factory: x2 / x1 unshuffle the input
    (PixelUnshuffle 2 / 4: conv1 sees 12 / 48 channels) and always run both nearest-x2 upsampling stages"""
    out = {}
    cases = [("x2_r2_s3", 2, 3.0, (2, 3, 16, 16)), ("x1_r2_s3", 1, 3.0, (1, 3, 16, 16)), ("x2_r2_s3_odd", 2, 3.0, (1, 3, 12, 20))]
    for name, s, scale, shape in cases:
        torch.manual_seed(0)
        g = MR.RRDBNet(in_channels=3, out_channels=3, channels=64, growth_channels=32, num_rrdb=2, upscale_factor=s)
        scaled_init(g, scale, 0.5)
        x = torch.rand(*shape)
        gt = torch.rand(shape[0], 3, shape[2] * s, shape[3] * s)
        sr = g(x)
        loss = torch.nn.functional.l1_loss(sr, gt)
        loss.backward()
        out[f"{name}/x"] = np_(x)
        out[f"{name}/gt"] = np_(gt)
        out[f"{name}/sr"] = np_(sr)
        out[f"{name}/loss"] = np.array(loss.item())
        named = dict(g.named_parameters())
        out[f"{name}/wsum"] = sd_checksums(g.state_dict())
        out[f"{name}/gsum"] = {k: checksum(p.grad) for k, p in named.items()}
        for k in ("conv1.weight", "conv1.bias", "conv4.weight", "trunk.0.rdb1.conv1.bias", "trunk.1.rdb3.conv5.bias", "conv2.bias"):
            out[f"{name}/grad/{k}"] = np_(named[k].grad)
    save("realesrgan_rrdbnet.npz", **out)


def gold_discriminator(MB):
    out = {}
    torch.manual_seed(0)
    d = MB.discriminator_unet(in_channels=3, out_channels=1, channels=64)
    x = torch.rand(2, 3, 64, 64)
    out["x"] = np_(x)
    out["wsum0"] = sd_checksums(d.state_dict())
    d.train()
    for it in range(3):
        logits = d(x)
        out[f"train{it}_logits"] = np_(logits)
        sd = d.state_dict()
        for layer in ("down_block1", "up_block1", "conv3"):
            out[f"train{it}_{layer}_u"] = np_(sd[f"{layer}.0.weight_u"])
            out[f"train{it}_{layer}_v"] = np_(sd[f"{layer}.0.weight_v"])
        out[f"train{it}_uvsum"] = {k: checksum(v) for k, v in sd.items() if k.endswith(("_u", "_v"))}
    loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, torch.ones_like(logits))
    loss.backward()
    out["bce_ones"] = np.array(loss.item())
    named = dict(d.named_parameters())
    out["gsum"] = {k: checksum(p.grad) for k, p in named.items()}
    out["grad/conv1.weight"] = np_(named["conv1.weight"].grad)
    out["grad/conv4.weight"] = np_(named["conv4.weight"].grad)
    out["grad/conv4.bias"] = np_(named["conv4.bias"].grad)
    out["grad/conv3.0.weight_orig"] = np_(named["conv3.0.weight_orig"].grad)
    d.eval()
    with torch.no_grad():
        out["eval_logits"] = np_(d(x))
    # input gradient (needed by the generator's adversarial term): dgrad-only pass
    d.train()
    xin = x.clone().requires_grad_(True)
    for p in d.parameters():
        p.requires_grad = False
    lg = d(xin)
    torch.nn.functional.binary_cross_entropy_with_logits(lg, torch.ones_like(lg)).backward()
    out["train3_dx"] = np_(xin.grad)
    out["train3_logits"] = np_(lg)
    save("discriminator.npz", **out)


def _gan_iterations(d, g, fname, *, g_lr, d_lr, pixel_w, adv_w, d_probe):
    """Two iterations of the reference GAN loop (BSRGAN/train_bsrgan.py:387-483; identical statements in
    A-ESRGAN/train_aesrgan.py:396-483) around reference modules, torch.optim.Adam, AveragedModel and an (inert on
    CPU) GradScaler; content loss stubbed to zeros (no VGG weights)."""
    from torch.optim.swa_utils import AveragedModel
    out = {}
    decay = 0.999
    ema = AveragedModel(g, avg_fn=lambda a, p, n: (1 - decay) * a + decay * p)
    d_opt = torch.optim.Adam(d.parameters(), d_lr, (0.9, 0.999), 1e-4, 0.0)
    g_opt = torch.optim.Adam(g.parameters(), g_lr, (0.9, 0.999), 1e-4, 0.0)
    bce = torch.nn.BCEWithLogitsLoss()
    l1 = torch.nn.L1Loss()
    pw, cw, aw = torch.Tensor([pixel_w]), torch.Tensor([1.0]), torch.Tensor([adv_w])
    d.train()
    g.train()
    B, h = 2, 16
    out["wsum_g0"] = sd_checksums(g.state_dict())
    out["wsum_d0"] = sd_checksums(d.state_dict())
    for it in range(2):
        lr = torch.rand(B, 3, h, h)
        gt = torch.rand(B, 3, 4 * h, 4 * h)
        out[f"it{it}_lr"] = np_(lr)
        out[f"it{it}_gt"] = np_(gt)
        real = torch.full([B, 1, 4 * h, 4 * h], 1.0)
        fake = torch.full([B, 1, 4 * h, 4 * h], 0.0)
        for p in d.parameters():
            p.requires_grad = True
        d.zero_grad(set_to_none=True)
        gt_output = d(gt)
        d_loss_hr = bce(gt_output, real)
        d_loss_hr.backward(retain_graph=True)
        sr = g(lr)
        sr_output = d(sr.detach().clone())
        d_loss_sr = bce(sr_output, fake)
        d_loss_sr.backward()
        d_loss = d_loss_hr + d_loss_sr
        d_opt.step()
        for p in d.parameters():
            p.requires_grad = False
        g.zero_grad(set_to_none=True)
        pixel = l1(sr, gt)
        content = torch.zeros(1, 5)
        adv = bce(d(sr), real)
        pixel = torch.sum(torch.mul(pw, pixel))
        content = torch.sum(torch.mul(cw, content))
        adv = torch.sum(torch.mul(aw, adv))
        g_loss = pixel + content + adv
        g_loss.backward()
        g_opt.step()
        ema.update_parameters(g)
        out[f"it{it}_scalars"] = np.array([d_loss.item(), pixel.item(), content.item(), adv.item(),
                                          torch.sigmoid(gt_output.detach()).mean().item(),
                                          torch.sigmoid(sr_output.detach()).mean().item()])
        out[f"it{it}_sr"] = np_(sr)
        out[f"it{it}_wsum_g"] = sd_checksums(g.state_dict())
        out[f"it{it}_wsum_d"] = sd_checksums(d.state_dict())
        out[f"it{it}_wsum_ema"] = sd_checksums(ema.state_dict())
        out[f"it{it}_g_conv4_bias"] = np_(g.conv4.bias)
        out[f"it{it}_d_probe"] = np_(dict(d.named_parameters())[d_probe])
    save(fname, **out)


def gold_realesrgan_gan_steps(MR):
    """Two iterations of Real_ESRGAN/train_realesrgan.py:407-466 around Real_ESRGAN/model.py's own modules: GENERATOR first
    (pixel + content on the USM-sharpened GT, adversarial through the frozen D), then the discriminator on the plain GT and the
    detached SR; realesrgan_config.py:138-151 hyper-parameters; content loss stubbed to zeros (no VGG weights)."""
    from torch.optim.swa_utils import AveragedModel
    torch.manual_seed(0)
    d = MR.discriminator_unet(in_channels=3, out_channels=1, channels=64)
    g = MR.rrdbnet_x4(in_channels=3, out_channels=3, channels=64, growth_channels=32, num_rrdb=2)
    scaled_init(g, 3.0, 0.5)
    out = {}
    decay = 0.999
    ema = AveragedModel(g, avg_fn=lambda a, p, n: (1 - decay) * a + decay * p)
    d_opt = torch.optim.Adam(d.parameters(), 1e-4, (0.9, 0.99), 1e-4, 0.0)
    g_opt = torch.optim.Adam(g.parameters(), 1e-4, (0.9, 0.99), 1e-4, 0.0)
    bce, l1 = torch.nn.BCEWithLogitsLoss(), torch.nn.L1Loss()
    pw, cw, aw = torch.Tensor([1.0]), torch.Tensor([0.1, 0.1, 1.0, 1.0, 1.0]), torch.Tensor([0.1])
    d.train()
    g.train()
    B, h = 2, 16
    out["wsum_g0"] = sd_checksums(g.state_dict())
    out["wsum_d0"] = sd_checksums(d.state_dict())
    for it in range(2):
        lr = torch.rand(B, 3, h, h)
        gt = torch.rand(B, 3, 4 * h, 4 * h)
        gt_usm = (gt + 0.1 * (gt - torch.nn.functional.avg_pool2d(gt, 3, 1, 1))).clamp(0, 1)       # any sharpened copy
        out[f"it{it}_lr"], out[f"it{it}_gt"], out[f"it{it}_gt_usm"] = np_(lr), np_(gt), np_(gt_usm)
        real = torch.full([B, 1, 4 * h, 4 * h], 1.0)
        fake = torch.full([B, 1, 4 * h, 4 * h], 0.0)
        for p in d.parameters():
            p.requires_grad = False
        g.zero_grad(set_to_none=True)
        sr = g(lr)
        pixel = l1(sr, gt_usm)
        feature = torch.zeros(1, 5)
        adv = bce(d(sr), real)
        pixel = torch.sum(torch.mul(pw, pixel))
        content = torch.sum(torch.mul(cw, feature))
        adv = torch.sum(torch.mul(aw, adv))
        (pixel + content + adv).backward()
        g_opt.step()
        for p in d.parameters():
            p.requires_grad = True
        d.zero_grad(set_to_none=True)
        gt_output = d(gt)
        d_loss_gt = bce(gt_output, real)
        d_loss_gt.backward()
        sr_output = d(sr.detach().clone())
        d_loss_sr = bce(sr_output, fake)
        d_loss_sr.backward()
        d_opt.step()
        ema.update_parameters(g)
        out[f"it{it}_scalars"] = np.array([(d_loss_sr + d_loss_gt).item(), pixel.item(), content.item(), adv.item(),
                                          torch.sigmoid_(torch.mean(gt_output.detach())).item(), torch.sigmoid_(torch.mean(sr_output.detach())).item()])
        out[f"it{it}_sr"] = np_(sr)
        out[f"it{it}_wsum_g"] = sd_checksums(g.state_dict())
        out[f"it{it}_wsum_d"] = sd_checksums(d.state_dict())
        out[f"it{it}_wsum_ema"] = sd_checksums(ema.state_dict())
        out[f"it{it}_g_conv4_bias"] = np_(g.conv4.bias)
        out[f"it{it}_d_probe"] = np_(dict(d.named_parameters())["conv4.weight"])
    save("realesrgan_gan_steps.npz", **out)


def gold_gan_steps(MB):
    torch.manual_seed(0)
    d = MB.discriminator_unet(in_channels=3, out_channels=1, channels=64)
    g = MB.bsrgan_x4(in_channels=3, out_channels=3, channels=64, growth_channels=32, num_rrdb=2)
    scaled_init(g, 3.0, 0.5)
    # bsrgan_config.py:137-151
    _gan_iterations(d, g, "gan_steps.npz", g_lr=8e-5, d_lr=2e-4, pixel_w=20.0, adv_w=0.5, d_probe="conv4.weight")


def gold_aesrgan_gan_steps(MA):
    """BASELINE config 5 pairing: RRDBNet x4 generator (A-ESRGAN/model.py:479-553) + attention U-Net discriminator."""
    torch.manual_seed(0)
    d = MA.uNetDiscriminatorAesrgan()
    g = MA.BSRGAN(in_channels=3, out_channels=3, channels=64, growth_channels=32, num_rrdb=2, upscale_factor=4)
    scaled_init(g, 3.0, 0.5)
    # aesrgan_config.py:137-155
    _gan_iterations(d, g, "aesrgan_gan_steps.npz", g_lr=5e-5, d_lr=1e-5, pixel_w=10.0, adv_w=0.1, d_probe="conv9.weight")


def gold_g_only_steps(MB, ME):
    """Two generator-only iterations: ESRGAN/train_rrdbnet.py:244-267 (Adam eps 1e-8, betas .9/.99, lr 2e-4)
    and BSRGAN/train_bsrnet.py:244-272 (eps 1e-4, lr 1e-4); cfg1 = BASELINE.json configs[0]."""
    out = {}
    cases = [
        ("esrgan_small", ME.rrdbnet_x4, dict(num_blocks=2), (2, 16), 2e-4, (0.9, 0.99), 1e-8, 3.0),
        ("bsrnet_small", MB.bsrgan_x4, dict(num_rrdb=2), (2, 16), 1e-4, (0.9, 0.99), 1e-4, 3.0),
        ("cfg1_esrgan_b4_32", ME.rrdbnet_x4, dict(num_blocks=23), (4, 32), 2e-4, (0.9, 0.99), 1e-8, 3.0),
    ]
    for name, fac, kw, (B, h), lr_, betas, eps, scale in cases:
        torch.manual_seed(0)
        g = fac(in_channels=3, out_channels=3, channels=64, growth_channels=32, **kw)
        scaled_init(g, scale, 0.5)
        opt = torch.optim.Adam(g.parameters(), lr_, betas, eps, 0.0)
        losses = []
        for it in range(2):
            lr = torch.rand(B, 3, h, h)
            gt = torch.rand(B, 3, 4 * h, 4 * h)
            if B * h * h <= 1024:
                out[f"{name}/it{it}_lr"] = np_(lr)
                out[f"{name}/it{it}_gt"] = np_(gt)
            g.zero_grad(set_to_none=True)
            sr = g(lr)
            loss = torch.mul(1.0, torch.nn.functional.l1_loss(sr, gt))
            loss.backward()
            opt.step()
            losses.append(loss.item())
            out[f"{name}/it{it}_wsum"] = sd_checksums(g.state_dict())
            if B * h * h <= 1024:
                out[f"{name}/it{it}_sr"] = np_(sr)
        out[f"{name}/losses"] = np.array(losses)
        out[f"{name}/conv4_bias"] = np_(g.conv4.bias)
    save("g_only_steps.npz", **out)


def gold_aesrgan_discriminator(MA):
    """UNetDiscriminatorAesrgan (A-ESRGAN/model.py:279-345): two training forwards (SN u/v + BatchNorm running
    stats advance), backward of BCE vs ones, eval forward, input gradient with frozen parameters."""
    out = {}
    torch.manual_seed(0)
    d = MA.UNetDiscriminatorAesrgan(3)
    x = torch.rand(2, 3, 64, 64)
    out["x"] = np_(x)
    out["wsum0"] = sd_checksums(d.state_dict())
    d.train()
    for it in range(2):
        logits = d(x)
        out[f"train{it}_logits"] = np_(logits)
        sd = d.state_dict()
        out[f"train{it}_statesum"] = {k: checksum(v) for k, v in sd.items()
                                      if k.endswith(("_u", "_v", "running_mean", "running_var"))}
        out[f"train{it}_attn3"] = np_(d.ly3)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, torch.ones_like(logits))
    loss.backward()
    out["bce_ones"] = np.array(loss.item())
    named = dict(d.named_parameters())
    out["gsum"] = {k: checksum(p.grad) for k, p in named.items()}
    for k in ("conv0.weight", "conv9.weight", "conv9.bias", "attn_1.W.1.weight", "attn_1.W.1.bias", "attn_3.psi.weight",
              "attn_2.theta.weight", "attn_3.phi.bias", "gating.weight_orig"):
        out[f"grad/{k}"] = np_(named[k].grad)
    d.eval()
    with torch.no_grad():
        out["eval_logits"] = np_(d(x))
    d.train()
    xin = x.clone().requires_grad_(True)
    for p in d.parameters():
        p.requires_grad = False
    lg = d(xin)
    torch.nn.functional.binary_cross_entropy_with_logits(lg, torch.ones_like(lg)).backward()
    out["train2_dx"] = np_(xin.grad)
    out["train2_logits"] = np_(lg)
    save("aesrgan_discriminator.npz", **out)


def gold_esrgan_discriminator(ME):
    """Discriminator (ESRGAN/model.py:88-141): two training forwards (BatchNorm running statistics advance), backward of
    BCE vs ones, eval forward, input gradient with frozen parameters.  BatchNorm affine parameters are randomised so
    that gamma / beta gradients and the normalisation are exercised away from (1, 0)."""
    out = {}
    torch.manual_seed(0)
    d = ME.discriminator()
    with torch.no_grad():
        for m in d.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.3, 0.3)
    x = torch.rand(4, 3, 128, 128)
    out["x"] = np_(x)
    out["wsum0"] = sd_checksums(d.state_dict())
    d.train()
    for it in range(2):
        logits = d(x)
        out[f"train{it}_logits"] = np_(logits)
        out[f"train{it}_statesum"] = {k: checksum(v) for k, v in d.state_dict().items() if k.endswith(("running_mean", "running_var"))}
    loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, torch.ones_like(logits))
    loss.backward()
    out["bce_ones"] = np.array(loss.item())
    named = dict(d.named_parameters())
    out["gsum"] = {k: checksum(p.grad) for k, p in named.items()}
    for k in ("features.0.weight", "features.0.bias", "features.3.weight", "features.3.bias", "features.27.weight", "classifier.0.bias",
              "classifier.2.weight", "classifier.2.bias"):
        out[f"grad/{k}"] = np_(named[k].grad)
    for k in ("features.2.weight", "features.14.weight", "features.26.weight", "classifier.0.weight"):   # first rows only (fixture size)
        out[f"gradrows/{k}"] = np_(named[k].grad[:2])
    d.eval()
    with torch.no_grad():
        out["eval_logits"] = np_(d(x))
    d.train()
    xin = x.clone().requires_grad_(True)
    for p in d.parameters():
        p.requires_grad = False
    lg = d(xin)
    torch.nn.functional.binary_cross_entropy_with_logits(lg, torch.ones_like(lg)).backward()
    out["train2_dx"] = np_(xin.grad)
    out["train2_logits"] = np_(lg)
    save("esrgan_discriminator.npz", **out)


def gold_esrgan_gan_steps(ME):
    """Two iterations of ESRGAN/train_esrgan.py:340-431 (generator first, relativistic-average losses, discriminator with
    three training forwards and retain_graph) around the reference's RRDBNet / Discriminator, torch.optim.Adam,
    AveragedModel; content loss stubbed to zero (no VGG weights); esrgan_config.py:95-111 hyper-parameters."""
    from torch.optim.swa_utils import AveragedModel
    out = {}
    torch.manual_seed(0)
    d = ME.discriminator()
    g = ME.rrdbnet_x4(in_channels=3, out_channels=3, channels=64, growth_channels=32, num_blocks=2)
    scaled_init(g, 3.0, 0.5)
    decay = 0.99998
    ema = AveragedModel(g, avg_fn=lambda a, p, n: (1 - decay) * a + decay * p)
    d_opt = torch.optim.Adam(d.parameters(), 1e-4, (0.9, 0.99), 1e-8, 0.0)
    g_opt = torch.optim.Adam(g.parameters(), 1e-4, (0.9, 0.99), 1e-8, 0.0)
    bce, l1 = torch.nn.BCEWithLogitsLoss(), torch.nn.L1Loss()
    d.train()
    g.train()
    B = 2
    out["wsum_g0"], out["wsum_d0"] = sd_checksums(g.state_dict()), sd_checksums(d.state_dict())
    for it in range(2):
        lr, gt = torch.rand(B, 3, 32, 32), torch.rand(B, 3, 128, 128)
        out[f"it{it}_lr"], out[f"it{it}_gt"] = np_(lr), np_(gt)
        real, fake = torch.full([B, 1], 1.0), torch.full([B, 1], 0.0)
        for p in d.parameters():
            p.requires_grad = False
        g.zero_grad(set_to_none=True)
        sr = g(lr)
        gt_output = d(gt.detach().clone())
        sr_output = d(sr)
        pixel = 0.01 * l1(sr, gt)
        content = 1.0 * torch.zeros(())
        adv = 0.005 * (bce(gt_output - torch.mean(sr_output), fake) * 0.5 + bce(sr_output - torch.mean(gt_output), real) * 0.5)
        (pixel + content + adv).backward()
        g_opt.step()
        ema.update_parameters(g)
        for p in d.parameters():
            p.requires_grad = True
        d.zero_grad(set_to_none=True)
        gt_output = d(gt)
        sr_output = d(sr.detach().clone())
        d_loss_gt = bce(gt_output - torch.mean(sr_output), real) * 0.5
        d_loss_gt.backward(retain_graph=True)
        sr_output = d(sr.detach().clone())
        d_loss_sr = bce(sr_output - torch.mean(gt_output), fake) * 0.5
        d_loss_sr.backward()
        d_opt.step()
        out[f"it{it}_scalars"] = np.array([(d_loss_gt + d_loss_sr).item(), pixel.item(), adv.item(),
                                          torch.sigmoid(torch.mean(gt_output.detach())).item(), torch.sigmoid(torch.mean(sr_output.detach())).item()])
        out[f"it{it}_sr"] = np_(sr[:, :, ::4, ::4])          # every 4th pixel (fixture size)
        out[f"it{it}_wsum_g"], out[f"it{it}_wsum_d"] = sd_checksums(g.state_dict()), sd_checksums(d.state_dict())
    save("esrgan_gan_steps.npz", **out)


def gold_validation():
    """Validation / data side: _psnr_torch (BSRGAN/image_quality_assessment.py:361-395) and random_crop
    (BSRGAN/imgproc.py:846-886), imported with an empty `cv2` stub module (neither function touches cv2)."""
    import random
    if "cv2" not in sys.modules:
        sys.modules["cv2"] = types.ModuleType("cv2")
    path = os.path.join(REF, "BSRGAN")
    sys.path.insert(0, path)
    try:
        for m in ("imgproc", "image_quality_assessment"):
            sys.modules.pop(m, None)
        imgproc = importlib.import_module("imgproc")
        iqa = importlib.import_module("image_quality_assessment")
    finally:
        sys.path.remove(path)
    out = {}
    torch.manual_seed(0)
    a = torch.rand(3, 3, 40, 56)
    b = (a + 0.05 * torch.randn_like(a)).clamp(0, 1)
    out["psnr_a"], out["psnr_b"] = np_(a), np_(b)
    out["psnr_y_cb4"] = np_(iqa._psnr_torch(a, b, 4, True))
    out["psnr_rgb_cb4"] = np_(iqa._psnr_torch(a, b, 4, False))
    out["psnr_y_cb0"] = np_(iqa.PSNR(0, True)(a, b))
    # SSIM (image_quality_assessment.py:420-494).  The SSIM module's constructor calls cv2.getGaussianKernel (OpenCV is not
    # in this image), so the module-level functions it forwards to are called directly with the window OpenCV documents:
    # G_i ~ exp(-(i-5)^2 / (2*1.5^2)), normalised, outer product (image_quality_assessment.py:520-521).
    xs = np.arange(11, dtype=np.float64) - 5.0
    gk = np.exp(-(xs ** 2) / (2.0 * 1.5 ** 2))
    gk = (gk / gk.sum()).reshape(11, 1)
    win = np.outer(gk, gk.transpose())
    out["ssim_window"] = win
    out["ssim_y_cb4"] = np_(iqa._ssim_single_torch(a, b, 4, True, 11, win))
    out["ssim_rgb_cb4"] = np_(iqa._ssim_single_torch(a, b, 4, False, 11, win))
    out["ssim_y_cb0"] = np_(iqa._ssim_single_torch(a, b, 0, True, 11, win))
    out["ssim_y_rolled"] = np_(iqa._ssim_single_torch(a, torch.roll(b, 3, dims=3), 0, True, 11, win))   # decorrelated pair
    box = np.full((7, 7), 1.0 / 49.0)
    out["ssim_box7_rgb_cb2"] = np_(iqa._ssim_single_torch(a, b, 2, False, 7, box))
    gt = torch.rand(2, 3, 48, 64)
    lr = torch.rand(2, 3, 12, 16)
    out["crop_gt"], out["crop_lr"] = np_(gt), np_(lr)
    for seed in (7, 11):
        random.seed(seed)
        pg, pl = imgproc.random_crop(gt, lr, 32, 4)
        out[f"crop{seed}_gt"], out[f"crop{seed}_lr"] = np_(pg), np_(pl)
    save("validation.npz", **out)


def gold_degradation():
    """Real-ESRGAN's on-device degradation stages: filter2d_torch, USMSharp.forward, DiffJPEG
    (Real_ESRGAN/imgproc.py:1092-1124, :1517-1540, :1465-1497), imported with empty stubs for cv2, torchvision and
    scipy's stats (none of the captured functions touches them).  USMSharp.__init__ calls cv2.getGaussianKernel, so the
    instance is assembled around the kernel OpenCV documents for (51, 0): sigma = 0.3*((51-1)*0.5-1)+0.8 = 8."""
    for name in ("cv2", "torchvision", "torchvision.transforms", "torchvision.transforms.functional",
                 "torchvision.transforms.functional_tensor"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision.transforms"].functional = sys.modules["torchvision.transforms.functional"]
    sys.modules["torchvision.transforms.functional_tensor"].rgb_to_grayscale = None
    path = os.path.join(REF, "Real_ESRGAN")
    sys.path.insert(0, path)
    try:
        sys.modules.pop("imgproc", None)
        ip = importlib.import_module("imgproc")
    finally:
        sys.path.remove(path)
        sys.modules.pop("imgproc", None)
    out = {}
    torch.manual_seed(0)
    img = torch.rand(3, 3, 45, 70)
    img[1] = torch.nn.functional.interpolate(torch.rand(1, 3, 12, 18), size=(45, 70), mode="bicubic").clamp(0, 1)[0]    # a smooth image too
    out["image"] = np_(img)
    k21 = torch.rand(3, 21, 21)
    k21 = k21 / k21.sum(dim=(1, 2), keepdim=True)
    out["kernels21"] = np_(k21)
    with torch.no_grad():
        out["filter2d_per_image"] = np_(ip.filter2d_torch(img, k21))
        out["filter2d_shared"] = np_(ip.filter2d_torch(img, k21[:1]))
        k7 = torch.rand(1, 7, 7)
        out["kernel7"] = np_(k7)
        out["filter2d_k7"] = np_(ip.filter2d_torch(img, k7))
        # USM sharpener on an image large enough for the 25-pixel reflect padding
        big = torch.nn.functional.interpolate(torch.rand(2, 3, 10, 12), size=(60, 72), mode="bicubic").clamp(0, 1)
        big = (big + 0.03 * torch.randn_like(big)).clamp(0, 1)
        out["usm_image"] = np_(big)
        xs = np.arange(51, dtype=np.float64) - 25.0
        g = np.exp(-(xs ** 2) / (2.0 * 8.0 ** 2))
        g = (g / g.sum()).reshape(51, 1)
        usm = ip.USMSharp.__new__(ip.USMSharp)
        torch.nn.Module.__init__(usm)
        usm.radius = 51
        usm.register_buffer("kernel", torch.FloatTensor(np.dot(g, g.transpose())).unsqueeze_(0))
        out["usm_kernel"] = np_(usm.kernel)
        out["usm_w05_t10"] = np_(usm(big, 0.5, 10))
        out["usm_w15_t3"] = np_(usm(big, 1.5, 3))
        # DiffJPEG: ragged size (45x70 -> padded to 48x80), per-image quality on both sides of 50, both roundings
        quality = torch.tensor([30.0, 75.0, 95.0])
        out["jpeg_quality"] = np_(quality)
        q = quality.clone()
        out["jpeg"] = np_(ip.DiffJPEG(False)(img, q))
        out["jpeg_factor"] = np_(q)                      # the reference converts the tensor it is given in place
        out["jpeg_diff"] = np_(ip.DiffJPEG(True)(img, quality.clone()))
        out["jpeg_scalar_q60"] = np_(ip.DiffJPEG(False)(img[:, :, :32, :48], 60))
        # noise stages: the reference's own functions under torch.manual_seed (CPU generator); the tests replay the draws
        sigma = torch.tensor([5.0, 20.0, 12.0])
        gray = torch.tensor([0.0, 1.0, 0.0])
        torch.manual_seed(123)
        out["gauss_gray"] = np_(ip._add_gaussian_noise_torch(img, sigma, True, False, gray))
        torch.manual_seed(124)
        out["gauss_color_rounds"] = np_(ip._add_gaussian_noise_torch(img, sigma, True, True, torch.zeros(3)))
        torch.manual_seed(125)
        out["gauss_random"] = np_(ip.random_add_gaussian_noise_torch(img, sigma_range=[1, 30], gray_prob=0.4, clip=True, rounds=False))
        img8 = torch.clamp((img * 255.0).round(), 0, 255) / 255.             # 8-bit image: the Poisson rates are then exact on any device
        img8[2] = (img8[2] * 255 // 8 * 8) / 255.                            # fewer distinct levels -> a different vals for this image
        out["poisson_image"] = np_(img8)
        torch.manual_seed(126)
        out["poisson_color"] = np_(ip._add_poisson_noise_torch(img8, torch.tensor([0.5, 2.0, 1.0]), True, False, 0))
        torch.manual_seed(127)
        out["poisson_random"] = np_(ip.random_add_poisson_noise_torch(img8, scale_range=[0.05, 3], gray_prob=0.0, clip=True, rounds=False))
        # the whole second-order pipeline on CPU, seeds chosen so that both noise stages are Gaussian (a Poisson stage's draw
        # count depends on the rates, so a 1e-7 difference upstream would desynchronise the generator) and both final orders occur.
        # usm_sharpener=None: the reference calls usm_sharpener(gt) without the weight / threshold its forward requires.
        P = dict(first_blur_probability=1.0, resize_probability1=[0.2, 0.7, 0.1], resize_range1=[0.15, 1.5], gray_noise_probability1=0.4,
                 gaussian_noise_probability1=0.5, noise_range1=[1, 30], poisson_scale_range1=[0.05, 3], jpeg_range1=[30, 95],
                 second_blur_probability=0.8, resize_probability2=[0.3, 0.4, 0.3], resize_range2=[0.3, 1.2], gray_noise_probability2=0.4,
                 gaussian_noise_probability2=0.5, noise_range2=[1, 25], poisson_scale_range2=[0.05, 2.5], jpeg_range2=[30, 95])
        import random
        gt = torch.nn.functional.interpolate(torch.rand(2, 3, 16, 16), size=(128, 128), mode="bicubic").clamp(0, 1)
        gt = (gt + 0.02 * torch.randn_like(gt)).clamp(0, 1)
        gk1 = torch.rand(2, 21, 21) ** 4
        gk1 = gk1 / gk1.sum(dim=(1, 2), keepdim=True)
        gk2 = torch.rand(2, 21, 21) ** 4
        gk2 = gk2 / gk2.sum(dim=(1, 2), keepdim=True)
        sk = torch.zeros(2, 21, 21)
        sk[:, 10, 10] = 1.2
        sk[:, 9:12, 9:12] -= 0.2 / 9
        out["pipe_gt"], out["pipe_k1"], out["pipe_k2"], out["pipe_sinc"] = np_(gt), np_(gk1), np_(gk2), np_(sk)
        orig_g, orig_p = ip.random_add_gaussian_noise_torch, ip.random_add_poisson_noise_torch
        jpeg_ref = ip.DiffJPEG(False)

        class ContiguousJPEG(torch.nn.Module):
            """the reference's DiffJPEG returns a permuted view; on the CPU its filter2d_torch then fails in .view(), so the
            operator handed to degradation_process makes the same values contiguous"""
            def forward(self, x, quality):
                return jpeg_ref(x, quality).contiguous()
        seeds = []
        for seed in range(400):
            trace = []
            ip.random_add_gaussian_noise_torch = lambda *a, **k: (trace.append("g"), orig_g(*a, **k))[1]
            ip.random_add_poisson_noise_torch = lambda *a, **k: (trace.append("p"), orig_p(*a, **k))[1]
            random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
            try:
                _, _, lr = ip.degradation_process(gt, gk1, gk2, sk, 4, P, ContiguousJPEG(), None)
            except TypeError:
                continue                                  # grey Poisson noise needs torchvision's rgb_to_grayscale (absent)
            if trace != ["g", "g"]:
                continue
            seeds.append(seed)
            out[f"pipe_lr_seed{seed}"] = np_(lr)
            if len(seeds) == 4:
                break
        ip.random_add_gaussian_noise_torch, ip.random_add_poisson_noise_torch = orig_g, orig_p
        out["pipe_seeds"] = np.array(seeds)
        # batch augmentation: random_crop_torch is pure slicing (the rotate / flip helpers call torchvision, absent here)
        aug_gt, aug_lr = torch.rand(2, 3, 48, 64), torch.rand(2, 3, 12, 16)
        out["aug_gt"], out["aug_lr"] = np_(aug_gt), np_(aug_lr)
        for seed in (3, 8):
            random.seed(seed)
            (c_usm, c_gt), c_lr = ip.random_crop_torch([aug_gt * 0.5, aug_gt], aug_lr, 32, 4)
            out[f"aug_crop{seed}_gt_usm"], out[f"aug_crop{seed}_gt"], out[f"aug_crop{seed}_lr"] = np_(c_usm), np_(c_gt), np_(c_lr)
    save("degradation.npz", **out)


def main():
    torch.set_num_threads(8)
    if "--only-validation" in sys.argv:
        return gold_validation()
    if "--only-degradation" in sys.argv:
        return gold_degradation()
    if "--only-realesrgan" in sys.argv:
        return gold_realesrgan_gan_steps(load_ref("Real_ESRGAN"))
    if "--only-realesrgan-rrdbnet" in sys.argv:
        return gold_realesrgan_rrdbnet(load_ref("Real_ESRGAN"))
    MB = load_ref("BSRGAN")
    ME = load_ref("ESRGAN")
    gold_blocks(MB)
    gold_generator(MB, ME)
    gold_discriminator(MB)
    gold_gan_steps(MB)
    gold_g_only_steps(MB, ME)
    MA = load_ref("A-ESRGAN")
    gold_aesrgan_discriminator(MA)
    gold_aesrgan_gan_steps(MA)
    gold_validation()
    gold_esrgan_discriminator(ME)
    gold_esrgan_gan_steps(ME)
    gold_degradation()
    MR = load_ref("Real_ESRGAN")
    gold_realesrgan_gan_steps(MR)
    gold_realesrgan_rrdbnet(MR)


if __name__ == "__main__":
    main()
