"""GPU parity of ESRGAN's BatchNorm discriminator against vectors captured from the reference
(ESRGAN/model.py:88-141)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.util import checksum, load_golden, rounded_weights, table

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a = a.detach().double().cpu() if torch.is_tensor(a) else torch.as_tensor(np.asarray(a)).double()
    b = b.detach().double().cpu() if torch.is_tensor(b) else torch.as_tensor(np.asarray(b)).double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def _build():
    from sr_gan_fd_amd import model as M
    torch.manual_seed(0)
    d = M.discriminator()
    with torch.no_grad():
        for m in d.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.3, 0.3)
    return d


def _rel_l2(a, b):
    a = a.detach().double().cpu() if torch.is_tensor(a) else torch.as_tensor(np.asarray(a)).double()
    b = b.detach().double().cpu() if torch.is_tensor(b) else torch.as_tensor(np.asarray(b)).double()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


@pytest.mark.parametrize("c", [64, 512])
def test_batchnorm_leakyrelu_fused(c):
    """srganfd_batchnorm_act_fwd / _bwd (BatchNorm2d + LeakyReLU(0.2), channel blocks of 256) vs torch on the same data"""
    from sr_gan_fd_amd import _abi as A
    torch.manual_seed(2)
    n, h, w = 4, 8, 12
    L, st = A.lib(), A.stream_ptr()
    bn = torch.nn.BatchNorm2d(c)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
    x = (torch.randn(n, c, h, w) * 2 + 1).requires_grad_(True)
    y = F.leaky_relu(bn(x), 0.2)
    dy = torch.randn_like(y)
    y.backward(dy)
    xa = x.detach().permute(0, 2, 3, 1).contiguous().cuda()
    ya, dxa = torch.empty_like(xa), torch.empty_like(xa)
    rm, rv = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    save, ws = torch.empty(4 * c, device="cuda"), torch.empty(2048 * 256 + 768, device="cuda")
    gam, bet = bn.weight.detach().cuda(), bn.bias.detach().cuda()
    A.check(L.srganfd_batchnorm_act_fwd(A.view(xa), A.view(ya), A.F32, n * h * w, c, gam.data_ptr(), bet.data_ptr(), rm.data_ptr(), rv.data_ptr(),
                                        0.1, 1e-5, 1, save.data_ptr(), ws.data_ptr(), 0.2, st))
    dg, db = torch.empty(c, device="cuda"), torch.empty(c, device="cuda")
    dya = dy.permute(0, 2, 3, 1).contiguous().cuda()
    A.check(L.srganfd_batchnorm_act_bwd(A.view(xa), A.view(dya), A.view(dxa), A.F32, n * h * w, c, gam.data_ptr(), save.data_ptr(), dg.data_ptr(),
                                        db.data_ptr(), 0.0, ws.data_ptr(), A.view(ya), 0.2, st))
    torch.cuda.synchronize()
    assert _rel(ya.permute(0, 3, 1, 2), y) < 1e-5
    assert _rel(rm, bn.running_mean) < 1e-5 and _rel(rv, bn.running_var) < 1e-5
    assert _rel(dxa.permute(0, 3, 1, 2), x.grad) < 1e-4
    assert _rel(dg, bn.weight.grad) < 1e-4 and _rel(db, bn.bias.grad) < 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_esrgan_discriminator(golden_dir, dtype):
    g = load_golden(golden_dir, "esrgan_discriminator.npz")
    f32 = dtype == torch.float32
    # 16-bit bounds per dtype (measured f16 / bf16: logits 1.8e-3 / 4.1e-2; sampled weight gradients, relative L2, worst 1.2e-1 / 2.5e-1 --
    # LeakyReLU masks of values within one 16-bit rounding of zero flip, each flip changes that element's gradient 5x; input gradient
    # 1.1e-1 / 2.2e-1): f16 is the benchmarked dtype and gets its own, tighter, numbers
    B16 = {torch.float16: dict(logits=5e-3, state=5e-3, loss=5e-3, grad=1.8e-1, dx=1.6e-1),
           torch.bfloat16: dict(logits=6e-2, state=3e-2, loss=5e-2, grad=3e-1, dx=3e-1)}.get(dtype)
    d = _build()
    d.compute_dtype = dtype
    d.cuda().train()
    x = torch.tensor(g["x"]).cuda()
    for it in range(2):
        logits = d(x)
        e = _rel(logits, g[f"train{it}_logits"])
        print(f"ESRGAN D {dtype} train fwd {it}: logits err {e:.2e}")
        assert e < (1e-3 if f32 else B16["logits"])
        sd = d.state_dict()
        for k, want in table(g, f"train{it}_statesum").items():
            tol = 1e-3 if f32 else B16["state"]
            assert np.allclose(checksum(sd[k]), want, rtol=tol, atol=tol * abs(want[1]) + 1e-7), f"state {k}: {checksum(sd[k])} vs {want}"
    assert int(d.features[3].num_batches_tracked) == 2
    loss = F.binary_cross_entropy_with_logits(logits, torch.ones_like(logits))
    assert abs(loss.item() - float(g["bce_ones"])) < (1e-3 if f32 else B16["loss"])
    S = 65536.0 if dtype == torch.float16 else 1.0       # f16 backward runs loss-scaled, as under the reference's GradScaler
    (loss * S).backward()
    named = dict(d.named_parameters())
    for p in named.values():
        p.grad /= S
    # The stack is conv -> BatchNorm -> LeakyReLU nine times: normalised values sit densely around zero, so a conv output
    # that differs from the reference's by fp32 summation order (~5e-6 here) flips a few dozen LeakyReLU masks in the
    # large early maps; each flip changes one element's gradient 5x and, through BatchNorm's channel sums, nudges its
    # whole channel.  (Measured: stage-9 gradients agree to 2e-6, one flipped element of 131072 in stage 8; the fused
    # BatchNorm+LeakyReLU kernels themselves are checked to 1e-4 above.)  Hence L2 bounds, looser towards the input.
    worst = 0.0
    for key in g.files:
        if key.startswith("grad/"):
            e = _rel_l2(named[key[5:]].grad, g[key])
        elif key.startswith("gradrows/"):
            e = _rel_l2(named[key[9:]].grad[:2], g[key])
        else:
            continue
        worst = max(worst, e)
        print(f"  {key}: L2 err {e:.2e}")
        deep = any(t in key for t in ("features.26", "features.27", "classifier"))
        assert e < ((1e-4 if deep else 2e-2) if f32 else B16["grad"]), f"{key}: {e:.2e}"
    print(f"ESRGAN D {dtype}: worst sampled grad L2 err {worst:.2e}")
    if f32:
        for k, want in table(g, "gsum").items():
            got = checksum(named[k].grad)
            assert np.allclose(got, want, rtol=2e-2, atol=5e-3 * abs(want[1]) + 1e-6), f"grad checksum {k}: {got} vs {want}"
    d.eval()
    with torch.no_grad():
        assert _rel(d(x), g["eval_logits"]) < (1e-3 if f32 else B16["logits"])
    d.train()
    for p in d.parameters():
        p.requires_grad = False
    xin = x.clone().requires_grad_(True)
    lg = d(xin)
    assert _rel(lg, g["train2_logits"]) < (1e-3 if f32 else B16["logits"])
    (F.binary_cross_entropy_with_logits(lg, torch.ones_like(lg)) * S).backward()
    ref = torch.tensor(g["train2_dx"]).double()
    e2 = ((xin.grad.double().cpu() / S - ref).norm() / ref.norm()).item()
    print(f"ESRGAN D {dtype}: input-gradient L2 err {e2:.2e}")
    assert e2 < (3e-2 if f32 else B16["dx"])


F16_GRAD_L2 = 1.5e-1      # d/dSR of the single-node content loss in f16 vs the f16-weight oracle, relative L2 (sign flips of |sr_f - gt_f| ~ f16 error)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_differentiable_single_node_content_loss(dtype):
    """ESRGAN ContentLoss (one node, kept in the autograd graph: ESRGAN/model.py:258-292): value and d/dSR vs the CPU oracle
    (seeded VGG-19 weights: the ImageNet file is a network download, parity of VALUES stays unpinned as for BSRGAN's).
    Asserted: value 1e-4 (f32), 2e-3 (f16, oracle on the same f16-rounded conv weights), 3e-2 (bf16); d/dSR L2 1e-2 / F16_GRAD_L2 / 4e-1."""
    from oracle import srgan_oracle as O
    from sr_gan_fd_amd import model as M
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    cl = M.ContentLoss("features.34", mean, std)
    cl.compute_dtype = dtype
    cl.cuda()
    torch.manual_seed(3)
    sr, gt = torch.rand(2, 3, 48, 32), torch.rand(2, 3, 48, 32)
    s = sr.clone().cuda().requires_grad_(True)
    loss = cl(s, gt.cuda())
    (2.5 * loss).backward()
    P = {"features." + k: v.detach().cpu() for k, v in cl.features.state_dict().items()}
    if dtype == torch.float16:
        P = rounded_weights(P, dtype)
    so = sr.clone().requires_grad_(True)
    want = O.content_loss_single(so, gt, P, "features.34", mean, std)
    (2.5 * want).backward()
    f32 = dtype == torch.float32
    e_l = abs(loss.item() - want.item()) / abs(want.item())
    e_g = _rel_l2(s.grad, so.grad)
    print(f"single-node content loss {dtype}: value {loss.item():.6f} vs {want.item():.6f} (rel {e_l:.2e}), dSR L2 err {e_g:.2e}, max {_rel(s.grad, so.grad):.2e}")
    assert loss.dim() == 0 and e_l < {torch.float32: 1e-4, torch.float16: 2e-3, torch.bfloat16: 3e-2}[dtype]
    # L2 bound: ~1M ReLU inputs, fp32 conv outputs that differ from the reference's by summation order (~5e-6) -> a handful
    # of mask flips, each worth sqrt(1/#active) of a layer's gradient (the oracle in fp64 vs fp32 shows none: 6e-7)
    # bf16: the tap gradient is sign(sr_f - gt_f); wherever |sr_f - gt_f| is below the bf16 error of 16 chained convs the sign
    # flips outright, so the bf16 gradient is only loosely tied to the fp32 one (the reference's fp16 autocast has the same trait)
    # f16 (11 bits): the same trait, an order of magnitude weaker
    assert e_g < {torch.float32: 1e-2, torch.float16: F16_GRAD_L2, torch.bfloat16: 4e-1}[dtype]
    with torch.no_grad():
        assert abs(cl(sr.cuda(), gt.cuda()).item() - loss.item()) < 1e-6 * abs(loss.item()) + 1e-9


def test_maxpool_relu_backward_and_l1_sign_kernels():
    from sr_gan_fd_amd import _abi as A
    torch.manual_seed(4)
    n, c, h, w = 2, 32, 8, 12
    L, st = A.lib(), A.stream_ptr()
    z = torch.randn(n, c, h, w, requires_grad=True)
    y = F.max_pool2d(F.relu(z), 2, 2)
    dy = torch.randn_like(y)
    y.backward(dy)
    xa = F.relu(z).detach().permute(0, 2, 3, 1).contiguous().cuda()
    dya = dy.permute(0, 2, 3, 1).contiguous().cuda()
    dxa = torch.empty_like(xa)
    A.check(L.srganfd_maxpool2_relu_bwd(A.view(xa), A.view(dya), A.view(dxa), A.F32, n, h, w, c, st))
    a, b = torch.randn(n, h, w, c, device="cuda"), torch.randn(n, h, w, c, device="cuda")
    b[0, 0, 0, :4] = a[0, 0, 0, :4]                        # exact ties: sign(0) = 0
    up = torch.tensor([1.5], device="cuda")
    out = torch.empty_like(a)
    A.check(L.srganfd_l1_grad_views(A.view(a), A.view(b), A.view(out), A.F32, n * h * w, c, up.data_ptr(), 0.25, st))
    torch.cuda.synchronize()
    assert torch.equal(dxa.permute(0, 3, 1, 2).cpu(), z.grad)
    assert torch.equal(out, 0.375 * torch.sign(a - b))


def test_esrgan_relativistic_iterations_on_dropin_modules(golden_dir):
    """The reference's own loop body (ESRGAN/train_esrgan.py:364-431: torch.optim.Adam, BCEWithLogitsLoss, L1Loss, autograd
    with retain_graph and three live discriminator forwards) over the drop-in RRDBNet / Discriminator modules."""
    from sr_gan_fd_amd import model as M
    from tests.util import scaled_init
    g = load_golden(golden_dir, "esrgan_gan_steps.npz")
    torch.manual_seed(0)
    d = M.discriminator()
    gen = M.rrdbnet_x4(in_channels=3, out_channels=3, channels=64, growth_channels=32, num_blocks=2)
    scaled_init(gen, 3.0, 0.5)
    d.compute_dtype = gen.compute_dtype = torch.float32
    d.cuda().train()
    gen.cuda().train()
    d_opt = torch.optim.Adam(d.parameters(), 1e-4, (0.9, 0.99), 1e-8, 0.0)
    g_opt = torch.optim.Adam(gen.parameters(), 1e-4, (0.9, 0.99), 1e-8, 0.0)
    bce, l1 = torch.nn.BCEWithLogitsLoss(), torch.nn.L1Loss()
    for it in range(2):
        lr, gt = torch.tensor(g[f"it{it}_lr"]).cuda(), torch.tensor(g[f"it{it}_gt"]).cuda()
        B = gt.shape[0]
        real, fake = torch.full([B, 1], 1.0, device="cuda"), torch.full([B, 1], 0.0, device="cuda")
        for p in d.parameters():
            p.requires_grad = False
        gen.zero_grad(set_to_none=True)
        sr = gen(lr)
        gt_output = d(gt.detach().clone())
        sr_output = d(sr)
        pixel = 0.01 * l1(sr, gt)
        adv = 0.005 * (bce(gt_output - torch.mean(sr_output), fake) * 0.5 + bce(sr_output - torch.mean(gt_output), real) * 0.5)
        (pixel + adv).backward()
        g_opt.step()
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
        got = [(d_loss_gt + d_loss_sr).item(), pixel.item(), adv.item(), torch.sigmoid(torch.mean(gt_output.detach())).item(),
               torch.sigmoid(torch.mean(sr_output.detach())).item()]
        want = g[f"it{it}_scalars"]
        print(f"ESRGAN relativistic it{it}: got {got} want {list(want)}")
        # iteration 1 starts from parameters that already took sign-like Adam steps (eps 1e-8): see the oracle test's note
        assert np.allclose(got, want, rtol=1e-3 if it == 0 else 2e-2, atol=1e-5)
        assert _rel(sr[:, :, ::4, ::4], g[f"it{it}_sr"]) < (1e-3 if it == 0 else 2e-2)
    assert int(d.features[3].num_batches_tracked) == 10          # five training forwards per iteration


def _esrgan_pair(dtype):
    from sr_gan_fd_amd import model as M
    from tests.util import scaled_init
    torch.manual_seed(0)
    d = M.discriminator()
    gen = M.rrdbnet_x4(in_channels=3, out_channels=3, channels=64, growth_channels=32, num_blocks=2)
    scaled_init(gen, 3.0, 0.5)
    d.compute_dtype = gen.compute_dtype = dtype
    return gen.cuda().train(), d.cuda().train()


def test_esrgan_fused_relativistic_trainer_vs_reference(golden_dir):
    """EsrganGanTrainer.step == two iterations of ESRGAN/train_esrgan.py:340-431 run on the reference's own modules (esrgan_gan_steps.npz:
    generator first, relativistic-average losses, five BatchNorm-advancing discriminator forwards, Adam, EMA; content loss stubbed to 0):
    logged scalars, SR, every parameter's checksum after both optimizer steps.  Iteration 1 starts from parameters that took sign-like
    Adam steps (eps 1e-8), hence its looser bound -- the same the module-level loop above and the CPU oracle meet."""
    from sr_gan_fd_amd.gan_esrgan import EsrganGanTrainer
    from tests.util import checksum, table
    g = load_golden(golden_dir, "esrgan_gan_steps.npz")
    gen, d = _esrgan_pair(torch.float32)
    tr = EsrganGanTrainer(gen, d, None)
    for it in range(2):
        s = tr.step(torch.tensor(g[f"it{it}_lr"]).cuda(), torch.tensor(g[f"it{it}_gt"]).cuda()).cpu().numpy()
        want = g[f"it{it}_scalars"]                     # d_loss, pixel, adversarial, sigmoid(mean D(gt)), sigmoid(mean D(sr))
        got = [s[0], s[1], s[3], s[4], s[5]]
        print(f"ESRGAN fused it{it}: got {got} want {list(want)}")
        tol = 1e-3 if it == 0 else 2e-2
        assert np.allclose(got, want, rtol=tol, atol=1e-5)
        assert _rel(tr.sr[:, :, ::4, ::4], g[f"it{it}_sr"]) < tol
        if it == 0:
            for sd, key in ((gen.state_dict(), "it0_wsum_g"), (d.state_dict(), "it0_wsum_d")):
                for k, want_c in table(g, key).items():
                    if "num_batches_tracked" in k:
                        assert int(sd[k]) == int(want_c[0]), k
                    else:
                        # Adam with eps 1e-8 moves every element by ~lr * sign(g) = 1e-4: an element whose gradient is within rounding of
                        # zero lands on the other side (2e-4 in a checksum).  Bound: 5e-3 of the tensor's absolute sum, or four such flips
                        # (the 32- / 64-element biases start at zero, so their whole content is this step)
                        assert np.allclose(checksum(sd[k]), want_c, rtol=5e-3, atol=max(5e-3 * abs(want_c[1]), 8e-4)), f"{key} {k}"
    assert int(d.features[3].num_batches_tracked) == 10          # five training forwards per iteration


def test_esrgan_fused_trainer_matches_module_level_loop_with_content_loss():
    """With the differentiable single-node VGG content loss switched on (seeded random extractor: the ImageNet weights are not in the
    reference tree), the fused trainer and the script's own autograd loop over the drop-in modules take the same two steps."""
    from sr_gan_fd_amd import model as M
    from sr_gan_fd_amd.gan_esrgan import EsrganGanTrainer
    MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    data = []
    torch.manual_seed(7)
    for _ in range(2):
        data.append((torch.rand(2, 3, 32, 32).cuda(), torch.rand(2, 3, 128, 128).cuda()))
    # fused
    gen, d = _esrgan_pair(torch.float32)
    torch.manual_seed(1)
    cl = M.content_loss("features.34", MEAN, STD)
    cl.compute_dtype = torch.float32
    cl.cuda()
    tr = EsrganGanTrainer(gen, d, cl, ema_decay=None)
    fused = [tr.step(lr, gt).cpu().numpy().copy() for lr, gt in data]
    # the script's loop on the modules (torch.optim.Adam, autograd)
    gen2, d2 = _esrgan_pair(torch.float32)
    torch.manual_seed(1)
    cl2 = M.content_loss("features.34", MEAN, STD)
    cl2.compute_dtype = torch.float32
    cl2.cuda()
    d_opt = torch.optim.Adam(d2.parameters(), 1e-4, (0.9, 0.99), 1e-8, 0.0)
    g_opt = torch.optim.Adam(gen2.parameters(), 1e-4, (0.9, 0.99), 1e-8, 0.0)
    bce, l1 = torch.nn.BCEWithLogitsLoss(), torch.nn.L1Loss()
    for it, (lr, gt) in enumerate(data):
        B = gt.shape[0]
        real, fake = torch.full([B, 1], 1.0, device="cuda"), torch.full([B, 1], 0.0, device="cuda")
        for p in d2.parameters():
            p.requires_grad = False
        gen2.zero_grad(set_to_none=True)
        sr = gen2(lr)
        gt_output = d2(gt.detach().clone())
        sr_output = d2(sr)
        pixel, content = 0.01 * l1(sr, gt), 1.0 * cl2(sr, gt)
        adv = 0.005 * (bce(gt_output - torch.mean(sr_output), fake) * 0.5 + bce(sr_output - torch.mean(gt_output), real) * 0.5)
        (pixel + content + adv).backward()
        g_opt.step()
        for p in d2.parameters():
            p.requires_grad = True
        d2.zero_grad(set_to_none=True)
        gt_output = d2(gt)
        sr_output = d2(sr.detach().clone())
        d_loss_gt = bce(gt_output - torch.mean(sr_output), real) * 0.5
        d_loss_gt.backward(retain_graph=True)
        sr_output = d2(sr.detach().clone())
        d_loss_sr = bce(sr_output - torch.mean(gt_output), fake) * 0.5
        d_loss_sr.backward()
        d_opt.step()
        want = [(d_loss_gt + d_loss_sr).item(), pixel.item(), content.item(), adv.item(), torch.sigmoid(torch.mean(gt_output.detach())).item(),
                torch.sigmoid(torch.mean(sr_output.detach())).item()]
        print(f"it{it}: fused {list(fused[it][:6])} loop {want}")
        assert np.allclose(fused[it][:6], want, rtol=1e-3 if it == 0 else 2e-2, atol=1e-5)
    assert _rel(gen.conv1.weight, gen2.conv1.weight) < 2e-2 and _rel(d.classifier[2].weight, d2.classifier[2].weight) < 2e-2


def test_esrgan_fused_trainer_f16_stays_close_to_f32():
    """The benchmark dtype: f16 activations / loss-scaled gradients against the f32 run of the same two iterations -- scalars within 5e-3
    relative (BatchNorm statistics over 2 images at 128x128 amplify one f16 rounding of the logits), no skipped optimizer step."""
    from sr_gan_fd_amd.gan_esrgan import EsrganGanTrainer
    torch.manual_seed(11)
    data = [(torch.rand(2, 3, 32, 32).cuda(), torch.rand(2, 3, 128, 128).cuda()) for _ in range(2)]
    outs = {}
    for dt in (torch.float32, torch.float16):
        gen, d = _esrgan_pair(dt)
        tr = EsrganGanTrainer(gen, d, None)
        outs[dt] = np.stack([tr.step(lr, gt).cpu().numpy()[:6].copy() for lr, gt in data])
        if dt == torch.float16:
            rep = tr.scaler.report()
            assert rep["enabled"] and rep["optimizer_steps"] == 4 and rep["skipped"] == 0, rep
    a, b = outs[torch.float32][0], outs[torch.float16][0]
    err = np.abs(a - b) / np.maximum(np.abs(a), 1e-6)
    print("f32", a, "f16", b, "rel", err)
    assert err[[0, 1, 3, 4, 5]].max() < 5e-3


def test_esrgan_fused_trainer_checkpoint_resume_is_bitwise():
    """EsrganGanTrainer.state_dict() / load_state_dict() (ESRGAN/train_esrgan.py:216-262 checkpoints g, d, both optimizers and the EMA
    model every epoch and resumes from them): a trainer rebuilt from fresh modules + the checkpoint takes bitwise the same third
    iteration as the one that kept running -- weights, BatchNorm running statistics, Adam moments and step, EMA, loss scale."""
    from sr_gan_fd_amd.gan_esrgan import EsrganGanTrainer
    torch.manual_seed(11)
    data = [(torch.rand(2, 3, 32, 32).cuda(), torch.rand(2, 3, 128, 128).cuda()) for _ in range(3)]
    gen, d = _esrgan_pair(torch.float16)
    tr = EsrganGanTrainer(gen, d, None)
    for lr, gt in data[:2]:
        tr.step(lr, gt)
    ckpt = tr.state_dict()
    assert set(ckpt) == {"g", "d", "scaler"} and set(ckpt["g"]) == {"state_dict", "optimizer", "ema_state_dict"}
    assert int(ckpt["g"]["ema_state_dict"]["n_averaged"]) == 2 and len(ckpt["d"]["optimizer"]["state"]) == len(list(d.parameters()))
    ckpt = {k: ({kk: (vv if not torch.is_tensor(vv) else vv.clone()) for kk, vv in v.items()} if k != "scaler" else dict(v)) for k, v in ckpt.items()}
    ckpt["g"]["state_dict"] = {k: v.clone() for k, v in ckpt["g"]["state_dict"].items()}
    ckpt["d"]["state_dict"] = {k: v.clone() for k, v in ckpt["d"]["state_dict"].items()}
    ckpt["scaler"]["scale"] = 4096.0                       # a value the fresh trainer would not start with
    tr.scaler.scale = 4096.0
    gen2, d2 = _esrgan_pair(torch.float16)
    with torch.no_grad():
        for p in list(gen2.parameters()) + list(d2.parameters()):
            p.add_(0.01)                                   # the checkpoint, not the seed, must provide the weights
    tr2 = EsrganGanTrainer(gen2, d2, None)
    tr2.load_state_dict(ckpt)
    assert tr2.scaler.scale == 4096.0 and tr2.g_opt.n_averaged == 2
    a = tr.step(*data[2]).clone()
    b = tr2.step(*data[2]).clone()
    assert torch.equal(a, b)
    assert torch.equal(tr.g_opt.flat, tr2.g_opt.flat) and torch.equal(tr.d_opt.flat, tr2.d_opt.flat)
    assert torch.equal(tr.g_opt.ema, tr2.g_opt.ema) and torch.equal(tr.g_opt.m, tr2.g_opt.m) and torch.equal(tr.d_opt.v, tr2.d_opt.v)
    for (k, x), (_, y) in zip(d.state_dict().items(), d2.state_dict().items()):
        assert torch.equal(x, y), k


def test_esrgan_iteration_at_the_config_crop_f16_vs_oracle():
    """esrgan_config.py:73-74: x4, LR crops of 32 x 32 (GT 128 x 128 -- the size the BatchNorm discriminator's classifier fixes), the
    23-RRDB generator.  One relativistic GAN iteration (train_esrgan.py:364-431, content term off) of the fused trainer in float16 -- the
    scripts' autocast dtype -- against the fp32 CPU oracle; batch 4 instead of 16 keeps the oracle at seconds.  The generator's dense
    blocks run through the dense-block launch (8 x 16 tiles) in both of its passes.  Asserted: pixel loss within 1e-3 relative
    (BASELINE.json's tolerance); the discriminator-side scalars within 5e-3 (the bound of the f16-against-f32 test above: BatchNorm
    statistics over four images amplify one f16 rounding of the logits); SR within 1e-3."""
    from oracle import srgan_oracle as O
    from sr_gan_fd_amd import model as M, ops
    from sr_gan_fd_amd.gan_esrgan import EsrganGanTrainer
    from tests.util import scaled_init, sd_to_params
    torch.manual_seed(9)
    lr_img, gt = torch.rand(4, 3, 32, 32), torch.rand(4, 3, 128, 128)
    torch.manual_seed(0)
    d = M.discriminator()
    gen = M.rrdbnet_x4(in_channels=3, out_channels=3, channels=64, growth_channels=32, num_blocks=23)
    scaled_init(gen, 3.0, 0.5)
    G = sd_to_params(gen.state_dict())
    D = {k: v.detach().clone() for k, v in d.state_dict().items()}
    g_opt = O.AdamState(G, O.g_param_names(G))
    d_opt = O.AdamState(D, [k for k in D if k.endswith((".weight", ".bias"))])
    out = O.esrgan_gan_step(G, D, g_opt, d_opt, lr_img, gt)
    d.compute_dtype = gen.compute_dtype = torch.float16
    gen, d = gen.cuda().train(), d.cuda().train()
    tr = EsrganGanTrainer(gen, d, None)
    s = tr.step(lr_img.cuda(), gt.cuda()).cpu().numpy()
    sp = tr.ge._last
    n_chain = len([a for a in sp.fw if type(a) is ops.DenseChain]) + len([it for it in sp.bw if it[0] == "chain"])
    assert n_chain == 6 * 23 or ops.DENSE_CHAIN == "0", n_chain
    got = [s[0], s[1], s[3], s[4], s[5]]
    want = [out[k] for k in ("d_loss", "pixel_loss", "adversarial_loss", "d_gt_probability", "d_sr_probability")]
    rel = [abs(a - b) / max(abs(b), 1e-6) for a, b in zip(got, want)]
    e_sr = _rel(tr.sr, out["sr"])
    print(f"f16 ESRGAN iteration at 32 -> 128, 23 RRDB: got {got} want {want} rel {[f'{r:.1e}' for r in rel]}; SR {e_sr:.2e}")
    assert rel[1] < 1e-3 and max(rel) < 5e-3 and e_sr < 1e-3
    rep = tr.scaler.report()
    assert rep["optimizer_steps"] == 2 and rep["skipped"] == 0, rep
