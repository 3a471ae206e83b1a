"""The drop-in modules under the reference's UNCHANGED precision contract.

The reference's train loops wrap every forward in ``amp.autocast()`` (fp16 on a GPU: train_bsrgan.py:415-427,450-457;
train_bsrnet.py:250-252) and drive the backward passes through ONE ``amp.GradScaler`` (train_bsrgan.py:109,420,430,436-437,
463,466-467); ``validate()`` runs outside autocast, in fp32 (train_bsrgan.py:563).  The mirror modules follow that by
themselves -- ``compute_dtype`` is never set in this file -- and must meet the reference's own vectors at north_star's
1e-3 (SR pixels and loss values) in the training loops, and be the exact-fp32 mode outside autocast.
"""
import numpy as np
import pytest
import torch
from torch import amp

from tests.util import checksum, load_golden, rounded_weights, scaled_init, sd_to_params, table

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a = torch.as_tensor(np.asarray(a)).double().cpu() if not torch.is_tensor(a) else a.detach().double().cpu()
    b = torch.as_tensor(np.asarray(b)).double().cpu() if not torch.is_tensor(b) else b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def _last_dtype(module):
    """dtype of the plan the module's last forward ran on (engine internals: test-only)"""
    from sr_gan_fd_amd.engine import _ENGINES
    return _ENGINES[module]._last.dt


def _gan_modules():
    from sr_gan_fd_amd import model as M
    torch.manual_seed(0)
    d = M.discriminator_unet(in_channels=3, out_channels=1, channels=64)
    gen = M.bsrgan_x4(in_channels=3, out_channels=3, channels=64, growth_channels=32, num_rrdb=2)
    scaled_init(gen, 3.0, 0.5)
    assert gen.compute_dtype is None and d.compute_dtype is None          # the default: follow autocast
    return gen.cuda().train(), d.cuda().train()


def test_reference_gan_loop_under_autocast_meets_the_golden_vectors(golden_dir):
    """train_bsrgan.py:387-483 as written -- autocast around the forwards, scaler.scale(loss).backward(), scaler.step / update
    after each optimizer, AveragedModel -- over the mirror modules, against two iterations of the reference itself
    (gan_steps.npz, captured on its fp32 CPU path): every logged scalar and the SR pixels within 1e-3."""
    from torch.optim.swa_utils import AveragedModel
    g = load_golden(golden_dir, "gan_steps.npz")
    gen, d = _gan_modules()
    ema = AveragedModel(gen, avg_fn=lambda a, p, n: (1 - 0.999) * a + 0.999 * p)
    d_opt = torch.optim.Adam(d.parameters(), 2e-4, (0.9, 0.999), 1e-4, 0.0)
    g_opt = torch.optim.Adam(gen.parameters(), 8e-5, (0.9, 0.999), 1e-4, 0.0)
    bce, l1 = torch.nn.BCEWithLogitsLoss(), torch.nn.L1Loss()
    scaler = amp.GradScaler("cuda")
    for it in range(2):
        lr, gt = torch.tensor(g[f"it{it}_lr"]).cuda(), torch.tensor(g[f"it{it}_gt"]).cuda()
        real, fake = torch.ones(2, 1, 64, 64, device="cuda"), torch.zeros(2, 1, 64, 64, device="cuda")
        for p in d.parameters():
            p.requires_grad = True
        d.zero_grad(set_to_none=True)
        with amp.autocast("cuda"):
            gt_output = d(gt)
            d_loss_hr = bce(gt_output, real)
        scaler.scale(d_loss_hr).backward(retain_graph=True)
        with amp.autocast("cuda"):
            sr = gen(lr)
            sr_output = d(sr.detach().clone())
            d_loss_sr = bce(sr_output, fake)
        scaler.scale(d_loss_sr).backward()
        scaler.step(d_opt)
        scaler.update()
        for p in d.parameters():
            p.requires_grad = False
        gen.zero_grad(set_to_none=True)
        with amp.autocast("cuda"):
            pixel = 20.0 * l1(sr, gt)
            adv = 0.5 * bce(d(sr), real)
            g_loss = pixel + adv
        scaler.scale(g_loss).backward()
        scaler.step(g_opt)
        scaler.update()
        ema.update_parameters(gen)
        assert _last_dtype(gen) == torch.float16 and _last_dtype(d) == torch.float16
        got = [(d_loss_hr + d_loss_sr).item(), pixel.item(), 0.0, adv.item(), torch.sigmoid(gt_output).mean().item(),
               torch.sigmoid(sr_output).mean().item()]
        want = g[f"it{it}_scalars"]
        err = max(abs(a - b) / max(abs(b), 1e-6) for a, b in zip(got, want) if b != 0.0)
        e_sr = _rel(sr, g[f"it{it}_sr"])
        print(f"autocast GAN loop it{it}: scalars worst rel {err:.2e}, SR err {e_sr:.2e}")
        assert err < 1e-3 and e_sr < 1e-3
        # parameters after the two Adam steps (observed 1-2e-3: Adam's g / (sqrt(v) + eps) amplifies f16 gradient rounding
        # where |g| ~ eps; the f32-mode test holds these at 1e-3)
        assert _rel(gen.conv4.bias, g[f"it{it}_g_conv4_bias"]) < 5e-3
        assert _rel(d.conv4.weight, g[f"it{it}_d_probe"]) < 5e-3
    assert scaler.get_scale() == 65536.0                       # no overflow, no skipped step
    assert int(ema.n_averaged) == 2


@pytest.mark.parametrize("name,fac,kw,B,h,lr,eps", [
    ("bsrnet_small", "bsrgan_x4", dict(num_rrdb=2), 2, 16, 1e-4, 1e-4),           # BSRGAN/bsrnet_config.py:86-96
    ("cfg1_esrgan_b4_32", "rrdbnet_x4", dict(num_blocks=23), 4, 32, 2e-4, 1e-8),  # BASELINE.json configs[0]
])
def test_reference_g_only_loop_under_autocast_meets_the_golden_vectors(golden_dir, name, fac, kw, B, h, lr, eps):
    """train_bsrnet.py:244-272 / train_rrdbnet.py:244-267 as written (autocast + GradScaler + torch.optim.Adam) over the mirror
    module: loss and SR of both iterations within 1e-3 of the reference's (g_only_steps.npz), incl. BASELINE configs[0]."""
    from sr_gan_fd_amd import model as M
    g = load_golden(golden_dir, "g_only_steps.npz")
    torch.manual_seed(0)
    net = getattr(M, fac)(in_channels=3, out_channels=3, channels=64, growth_channels=32, **kw)
    scaled_init(net, 3.0, 0.5)
    stored = f"{name}/it0_lr" in g.files
    draws = [(torch.rand(B, 3, h, h), torch.rand(B, 3, 4 * h, 4 * h)) for _ in range(2)]
    net.cuda().train()
    opt = torch.optim.Adam(net.parameters(), lr, (0.9, 0.99), eps, 0.0)
    scaler = amp.GradScaler("cuda")
    crit = torch.nn.L1Loss()
    for it in range(2):
        x, gt = draws[it][0].cuda(), draws[it][1].cuda()
        net.zero_grad(set_to_none=True)
        with amp.autocast("cuda"):
            sr = net(x)
            loss = torch.mul(1.0, crit(sr, gt))
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        assert _last_dtype(net) == torch.float16
        want = float(g[f"{name}/losses"][it])
        print(f"autocast {name} it{it}: loss {loss.item():.7f} (reference {want:.7f})")
        assert abs(loss.item() - want) < 1e-3 * abs(want)
        if stored:
            assert _rel(sr, g[f"{name}/it{it}_sr"]) < 1e-3


def test_forward_outside_autocast_is_the_f32_mode(golden_dir):
    """validate() (train_bsrgan.py:563) calls the generator outside autocast: the mirror then computes in exact fp32 -- bit-equal to
    a module pinned with compute_dtype = float32, and within 1e-3 (observed 1e-6) of the reference's SR; the same call inside
    ``autocast(dtype=bfloat16)`` runs bf16, and an explicit compute_dtype wins over autocast."""
    from sr_gan_fd_amd import model as M
    g = load_golden(golden_dir, "generator.npz")

    def build():
        torch.manual_seed(0)
        net = M.bsrgan_x4(in_channels=3, out_channels=3, channels=64, growth_channels=32, num_rrdb=2)
        scaled_init(net, 3.0, 0.5)
        return net.cuda().eval()
    x = torch.tensor(g["bsrgan_x4_r2_s3/x"]).cuda()
    follow, pinned = build(), build()
    pinned.compute_dtype = torch.float32
    with torch.no_grad():
        a, b = follow(x), pinned(x)
        assert _last_dtype(follow) == torch.float32
        assert torch.equal(a, b)
        assert _rel(a, g["bsrgan_x4_r2_s3/sr"]) < 1e-5
        with amp.autocast("cuda"):
            h = follow(x)
            assert _last_dtype(follow) == torch.float16
            p = pinned(x)
            assert _last_dtype(pinned) == torch.float32 and torch.equal(p, b)
        with amp.autocast("cuda", dtype=torch.bfloat16):
            follow(x)
            assert _last_dtype(follow) == torch.bfloat16
        with amp.autocast("cuda", enabled=False):
            assert torch.equal(follow(x), b)
    assert h.dtype == torch.float32 and _rel(h, g["bsrgan_x4_r2_s3/sr"]) < 1e-3


def test_discriminator_and_content_loss_follow_autocast():
    """outside autocast fp32, inside it f16 -- and each result is compared with the CPU ORACLE (not with the other HIP run): the fp32
    forwards at 1e-3 / 1e-4, the f16 ones against the oracle on f16-rounded conv weights at 5e-3 (logits, max error over the scale)
    and 2e-3 (the (1, 5) content values)"""
    from oracle import srgan_oracle as O
    from sr_gan_fd_amd import model as M
    torch.manual_seed(0)
    d = M.discriminator_unet(in_channels=3, out_channels=1, channels=64).cuda().train()
    nodes, mean, std = ["features.2", "features.7", "features.16", "features.25", "features.34"], [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    cl = M.content_loss(feature_model_extractor_nodes=nodes, feature_model_normalize_mean=mean, feature_model_normalize_std=std).cuda().eval()
    x, y = torch.rand(2, 3, 64, 64, device="cuda"), torch.rand(2, 3, 64, 64, device="cuda")
    with torch.no_grad():
        for _ in range(3):
            d(x)                  # training-mode forwards: power iterations bring the freshly drawn u / v to a usable sigma
        d.eval()                  # eval: no iteration, both forwards below see the same normalised weights
        o32, c32 = d(x), cl(x, y)
        assert _last_dtype(d) == torch.float32 and _last_dtype(cl) == torch.float32
        with amp.autocast("cuda"):
            o16, c16 = d(x), cl(x, y)
        assert _last_dtype(d) == torch.float16 and _last_dtype(cl) == torch.float16
    assert o16.dtype == torch.float32 and tuple(c16.shape) == (1, 5)
    PD = {k: v.detach().cpu().clone() for k, v in d.state_dict().items()}
    PV = {"features." + k: v.detach().cpu() for k, v in cl.features.state_dict().items()}
    xc, yc = x.cpu(), y.cpu()
    with torch.no_grad():
        want_o32 = O.discriminator_unet_forward(xc, PD, training=False)
        want_c32 = O.content_loss(xc, yc, PV, nodes, mean, std, taps_post_relu=True)
        want_c16 = O.content_loss(xc, yc, rounded_weights(PV, torch.float16), nodes, mean, std, taps_post_relu=True)
    e = dict(o32=_rel(o32, want_o32), o16=_rel(o16, want_o32), c32=_rel(c32, want_c32), c16=_rel(c16, want_c16))
    print("D logits / content values vs the CPU oracle:", {k: f"{v:.2e}" for k, v in e.items()})
    assert e["o32"] < 1e-3 and e["c32"] < 1e-3
    # the discriminator's eval weights are weight_orig / sigma rounded to f16 by the packer: the fp32 oracle is the reference here
    assert e["o16"] < 5e-3 and e["c16"] < 2e-3


def test_fused_trainer_over_default_modules_trains_in_the_loops_float16():
    """a fused trainer is the reference's training loop, which runs under amp.autocast() + GradScaler: built over modules left at the
    default (compute_dtype None), inside OR outside an autocast region, it pins them to float16 with the loss scaler on (ADVICE r4:
    before, a trainer built outside autocast silently trained in the fp32 parity mode); a module pinned to float32 first keeps it, and
    entering a float16 autocast region afterwards is refused (its scaler is off)"""
    from sr_gan_fd_amd import _abi as A
    from sr_gan_fd_amd.trainer import GeneratorTrainer
    from sr_gan_fd_amd import model as M
    torch.manual_seed(0)
    x, gt = torch.rand(1, 3, 16, 16, device="cuda"), torch.rand(1, 3, 64, 64, device="cuda")
    net = M.bsrgan_x4(num_rrdb=1).cuda().train()
    tr = GeneratorTrainer(net, lr=1e-4)
    assert net.compute_dtype == torch.float16 and tr.scaler.enabled
    tr.step(x, gt)
    assert _last_dtype(net) == torch.float16
    with amp.autocast("cuda"):
        tr.step(x, gt)
        tr16 = GeneratorTrainer(M.bsrgan_x4(num_rrdb=1).cuda().train(), lr=1e-4)     # built inside the region: the same
        assert tr16.scaler.enabled
        tr16.step(x, gt)
    net32 = M.bsrgan_x4(num_rrdb=1).cuda().train()
    net32.compute_dtype = torch.float32
    tr32 = GeneratorTrainer(net32, lr=1e-4)
    assert not tr32.scaler.enabled
    tr32.step(x, gt)
    net32.compute_dtype = None               # back to "follow autocast" behind the trainer's back
    with amp.autocast("cuda"):
        with pytest.raises(A.SrganfdError, match="loss scaler is disabled"):
            tr32.step(x, gt)
