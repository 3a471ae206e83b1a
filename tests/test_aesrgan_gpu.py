"""GPU parity of the A-ESRGAN attention U-Net discriminator (BASELINE config 5 discriminator) and the kernels
it adds (stride-2 3x3 / 2x2 convs, padded 1x1 conv, general bilinear resize, BatchNorm, attention gate)
against vectors captured from the reference (A-ESRGAN/model.py:228-345)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.util import checksum, load_golden, scaled_init, table

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a = a.detach().double().cpu() if torch.is_tensor(a) else torch.as_tensor(np.asarray(a)).double()
    b = b.detach().double().cpu() if torch.is_tensor(b) else torch.as_tensor(np.asarray(b)).double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def _rel_l2(a, b):
    a = a.detach().double().cpu() if torch.is_tensor(a) else torch.as_tensor(np.asarray(a)).double()
    b = b.detach().double().cpu() if torch.is_tensor(b) else torch.as_tensor(np.asarray(b)).double()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


@pytest.mark.parametrize("sizes", [((10, 10), (8, 8)), ((10, 10), (32, 32)), ((6, 10), (16, 24)), ((8, 8), (16, 16))])
def test_resize_bilinear(sizes):
    from sr_gan_fd_amd import _abi as A
    (hi, wi), (ho, wo) = sizes
    torch.manual_seed(0)
    n, c = 2, 32
    x = torch.randn(n, c, hi, wi, requires_grad=True)
    y = F.interpolate(x, size=(ho, wo), mode="bilinear", align_corners=False)
    dy = torch.randn_like(y)
    y.backward(dy)
    L, st = A.lib(), A.stream_ptr()
    xa = x.detach().permute(0, 2, 3, 1).contiguous().cuda()
    yb = torch.empty(n, ho, wo, c, device="cuda")
    A.check(L.srganfd_resize_bilinear(0, A.view(xa), A.view(yb), A.F32, n, hi, wi, ho, wo, c, st))
    dya = dy.permute(0, 2, 3, 1).contiguous().cuda()
    dxb = torch.empty(n, hi, wi, c, device="cuda")
    A.check(L.srganfd_resize_bilinear(1, A.view(dya), A.view(dxb), A.F32, n, hi, wi, ho, wo, c, st))
    torch.cuda.synchronize()
    assert _rel(yb.permute(0, 3, 1, 2), y) < 1e-5
    assert _rel(dxb.permute(0, 3, 1, 2), x.grad) < 1e-5


def test_batchnorm_and_gate():
    from sr_gan_fd_amd import _abi as A
    torch.manual_seed(1)
    n, c, h, w = 2, 64, 12, 20
    L, st = A.lib(), A.stream_ptr()
    bn = torch.nn.BatchNorm2d(c)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
    x = torch.randn(n, c, h, w, requires_grad=True)
    y = bn(x)
    dy = torch.randn_like(y)
    y.backward(dy)
    xa = x.detach().permute(0, 2, 3, 1).contiguous().cuda()
    yb, dxb = torch.empty_like(xa), torch.empty_like(xa)
    rm, rv = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    save, ws = torch.empty(4 * c, device="cuda"), torch.empty(2048 * 256 + 768, device="cuda")
    gam, bet = bn.weight.detach().cuda(), bn.bias.detach().cuda()
    A.check(L.srganfd_batchnorm_fwd(A.view(xa), A.view(yb), A.F32, n * h * w, c, gam.data_ptr(), bet.data_ptr(), rm.data_ptr(), rv.data_ptr(),
                                    0.1, 1e-5, 1, save.data_ptr(), ws.data_ptr(), st))
    dg, db = torch.empty(c, device="cuda"), torch.empty(c, device="cuda")
    dya = dy.permute(0, 2, 3, 1).contiguous().cuda()
    A.check(L.srganfd_batchnorm_bwd(A.view(xa), A.view(dya), A.view(dxb), A.F32, n * h * w, c, gam.data_ptr(), save.data_ptr(), dg.data_ptr(),
                                    db.data_ptr(), 0.0, ws.data_ptr(), st))
    torch.cuda.synchronize()
    assert _rel(yb.permute(0, 3, 1, 2), y) < 1e-5
    assert _rel(rm, bn.running_mean) < 1e-5 and _rel(rv, bn.running_var) < 1e-5
    assert _rel(dxb.permute(0, 3, 1, 2), x.grad) < 1e-4
    assert _rel(dg, bn.weight.grad) < 1e-4 and _rel(db, bn.bias.grad) < 1e-4
    # gate
    xg = torch.randn(n, c, h, w, requires_grad=True)
    g = torch.rand(n, 1, h, w, requires_grad=True)
    yy = g.expand_as(xg) * xg
    yy.backward(dy)
    xa = xg.detach().permute(0, 2, 3, 1).contiguous().cuda()
    ga = g.detach().reshape(-1).cuda()
    yb = torch.empty_like(xa)
    A.check(L.srganfd_gate_mul(0, A.view(xa), ga.data_ptr(), A.view(yb), A.NULL_VIEW, None, A.F32, n * h * w, c, st))
    dgate = torch.empty(n * h * w, device="cuda")
    A.check(L.srganfd_gate_mul(1, A.view(xa), ga.data_ptr(), A.view(dya), A.view(dxb), dgate.data_ptr(), A.F32, n * h * w, c, st))
    torch.cuda.synchronize()
    assert _rel(yb.permute(0, 3, 1, 2), yy) < 1e-6
    assert _rel(dxb.permute(0, 3, 1, 2), xg.grad) < 1e-6
    assert _rel(dgate.view(n, 1, h, w), g.grad) < 1e-5


def test_sync_batchnorm_two_rank_emulation():
    """Two ranks holding half the batch each, the all-reduce emulated by adding their partial tables: the two-phase kernels
    (srganfd_batchnorm_{fwd,bwd}_sync) reproduce the full-batch output, running statistics and dx on both halves, and the
    two ranks' dgamma/dbeta add up to the full-batch parameter gradients (what the gradient all-reduce then sums)."""
    from sr_gan_fd_amd import _abi as A
    torch.manual_seed(3)
    n, c, h, w = 4, 64, 12, 20
    L, st = A.lib(), A.stream_ptr()
    bn = torch.nn.BatchNorm2d(c)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
    x = (torch.randn(n, c, h, w) * 2 + 0.7).requires_grad_(True)
    y = bn(x)
    dy = torch.randn_like(y)
    y.backward(dy)
    nfl = L.srganfd_batchnorm_partial_floats(c)
    assert nfl == 2 * 1024 * c
    gam, bet = bn.weight.detach().cuda(), bn.bias.detach().cuda()
    half = n // 2
    npix = half * h * w
    ranks = []
    for r in range(2):
        xa = x.detach()[r * half:(r + 1) * half].permute(0, 2, 3, 1).contiguous().cuda()
        dya = dy[r * half:(r + 1) * half].permute(0, 2, 3, 1).contiguous().cuda()
        ranks.append(dict(x=xa, dy=dya, y=torch.empty_like(xa), dx=torch.empty_like(xa), rm=torch.zeros(c, device="cuda"), rv=torch.ones(c, device="cuda"),
                          save=torch.empty(4 * c, device="cuda"), ws=torch.empty(2048 * 256 + 768, device="cuda"), wsg=torch.empty(nfl, device="cuda"),
                          dg=torch.empty(c, device="cuda"), db=torch.empty(c, device="cuda")))

    def fwd(R, phase, total):
        A.check(L.srganfd_batchnorm_fwd_sync(A.view(R["x"]), A.view(R["y"]), A.F32, npix, c, gam.data_ptr(), bet.data_ptr(), R["rm"].data_ptr(), R["rv"].data_ptr(),
                                             0.1, 1e-5, R["save"].data_ptr(), R["ws"].data_ptr(), 1.0, phase, total, st))

    def bwd(R, phase, total):
        A.check(L.srganfd_batchnorm_bwd_sync(A.view(R["x"]), A.view(R["dy"]), A.view(R["dx"]), A.F32, npix, c, gam.data_ptr(), R["save"].data_ptr(), R["dg"].data_ptr(),
                                             R["db"].data_ptr(), 0.0, R["ws"].data_ptr(), R["wsg"].data_ptr(), A.NULL_VIEW, 1.0, phase, total, st))

    for R in ranks:
        fwd(R, 1, 0)
    tot = ranks[0]["ws"][:nfl] + ranks[1]["ws"][:nfl]          # the all-reduce
    for R in ranks:
        R["ws"][:nfl] = tot
        fwd(R, 2, 2 * npix)
    for R in ranks:
        bwd(R, 1, 0)
    tot = ranks[0]["ws"][:nfl] + ranks[1]["ws"][:nfl]
    for R in ranks:
        R["wsg"].copy_(tot)
        bwd(R, 2, 2 * npix)
    torch.cuda.synchronize()
    yy = torch.cat([R["y"] for R in ranks]).permute(0, 3, 1, 2)
    dxx = torch.cat([R["dx"] for R in ranks]).permute(0, 3, 1, 2)
    assert _rel(yy, y) < 1e-5
    assert _rel(dxx, x.grad) < 1e-4
    for R in ranks:
        assert _rel(R["rm"], bn.running_mean) < 1e-5 and _rel(R["rv"], bn.running_var) < 1e-5
    assert _rel(ranks[0]["dg"] + ranks[1]["dg"], bn.weight.grad) < 1e-4
    assert _rel(ranks[0]["db"] + ranks[1]["db"], bn.bias.grad) < 1e-4
    # and the statistics differ from what each half alone would give (the test would pass vacuously otherwise)
    own = x.detach()[:half].mean(dim=(0, 2, 3))
    assert (own - x.detach().mean(dim=(0, 2, 3))).abs().max() > 1e-3
    # single-rank group: the two-phase path of the engine equals the single-call path
    with pytest.raises(A.SrganfdError):
        A.check(L.srganfd_batchnorm_bwd_sync(A.view(ranks[0]["x"]), A.view(ranks[0]["dy"]), A.view(ranks[0]["dx"]), A.F32, npix, c, gam.data_ptr(),
                                             ranks[0]["save"].data_ptr(), ranks[0]["dg"].data_ptr(), ranks[0]["db"].data_ptr(), 0.0, ranks[0]["ws"].data_ptr(),
                                             None, A.NULL_VIEW, 1.0, 2, 2 * npix, st))


def test_sync_batchnorm_engine_path_world_one():
    """The discriminator engine with the two-phase BatchNorm path on a one-rank group gives the single-call results"""
    from sr_gan_fd_amd import model as M
    from sr_gan_fd_amd.engine_a import aesrgan_engine
    from sr_gan_fd_amd.parallel import SyncBatchNormReduce
    torch.manual_seed(0)
    d = M.UNetDiscriminatorAesrgan(3).cuda().train()
    x = torch.rand(2, 3, 64, 64, device="cuda")
    e = aesrgan_engine(d)
    import copy
    sd = copy.deepcopy(d.state_dict())      # spectral-norm u/v and the running statistics move with every training forward
    outs = []
    for sync in (None, SyncBatchNormReduce(None)):
        e.sync_bn = sync
        d.load_state_dict(sd)
        for p in d.parameters():
            p.grad = None
        y = d(x)
        y.mean().backward()
        outs.append((y.detach().clone(), torch.cat([p.grad.flatten() for p in d.parameters()]).clone(),
                     torch.cat([m.running_var for m in d.modules() if isinstance(m, torch.nn.BatchNorm2d)]).clone()))
    e.sync_bn = None
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_aesrgan_discriminator(golden_dir, dtype):
    from sr_gan_fd_amd import model as M
    g = load_golden(golden_dir, "aesrgan_discriminator.npz")
    f32 = dtype == torch.float32
    # 16-bit bounds per dtype (measured f16 / bf16: logits 1.4e-3 / 1.0e-2, attention map 4e-5 / 4e-4, sampled weight gradients 3.5e-2 /
    # 1.7e-1, input gradient relative L2 5.2e-2 / 1.4e-1): float16 is the benchmarked dtype and gets its own, tighter, numbers
    B16 = {torch.float16: dict(logits=5e-3, grad=8e-2, dx=1e-1), torch.bfloat16: dict(logits=5e-2, grad=2e-1, dx=1.5e-1)}.get(dtype)
    torch.manual_seed(0)
    d = M.UNetDiscriminatorAesrgan(3)
    d.compute_dtype = dtype
    d.cuda().train()
    x = torch.tensor(g["x"]).cuda()
    for it in range(2):
        logits = d(x)
        e = _rel(logits, g[f"train{it}_logits"])
        print(f"A-ESRGAN D {dtype} train fwd {it}: logits err {e:.2e}, attention-map err {_rel(d.ly3, g[f'train{it}_attn3']):.2e}")
        assert e < (1e-3 if f32 else B16["logits"])
        assert _rel(d.ly3, g[f"train{it}_attn3"]) < (1e-3 if f32 else B16["logits"])
        sd = d.state_dict()
        for k, want in table(g, f"train{it}_statesum").items():
            tol = 1e-3 if (f32 or k.endswith(("_u", "_v"))) else 3e-2     # BN statistics see bf16 activations
            assert np.allclose(checksum(sd[k]), want, rtol=tol, atol=tol * abs(want[1]) + 1e-7), f"state {k}: {checksum(sd[k])} vs {want}"
    assert int(d.attn_1.W[1].num_batches_tracked) == 2
    loss = F.binary_cross_entropy_with_logits(logits, torch.ones_like(logits))
    assert abs(loss.item() - float(g["bce_ones"])) < (1e-3 if f32 else B16["logits"])
    S = 65536.0 if dtype == torch.float16 else 1.0       # f16 backward runs loss-scaled, as under the reference's GradScaler
    (loss * S).backward()
    named = dict(d.named_parameters())
    worst = 0.0
    for k in ("conv0.weight", "conv9.weight", "conv9.bias", "attn_1.W.1.weight", "attn_1.W.1.bias", "attn_3.psi.weight",
              "attn_2.theta.weight", "attn_3.phi.bias", "gating.weight_orig"):
        e = _rel(named[k].grad / S, g[f"grad/{k}"])
        worst = max(worst, e)
        assert e < (2e-3 if f32 else B16["grad"]), f"grad {k}: {e:.2e}"
    print(f"A-ESRGAN D {dtype}: worst sampled grad err {worst:.2e}")
    if f32:
        for k, want in table(g, "gsum").items():
            got = checksum(named[k].grad)
            assert np.allclose(got, want, rtol=2e-2, atol=2e-3 * abs(want[1]) + 1e-6), f"grad checksum {k}: {got} vs {want}"
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
    dxin = xin.grad / S
    e, e2 = _rel(dxin, g["train2_dx"]), _rel_l2(dxin, g["train2_dx"])
    ref = torch.as_tensor(np.asarray(g["train2_dx"])).double()
    err = (dxin.detach().double().cpu() - ref).abs() / ref.abs().max()
    frac = (err > 2e-3).double().mean().item()
    print(f"A-ESRGAN D {dtype}: input-gradient max err {e:.2e}, L2 err {e2:.2e}, fraction above 2e-3: {frac:.2e}")
    if f32:
        # The ReLU / LeakyReLU masks are discontinuous: a pre-activation that lands within one fp32 rounding of zero
        # takes the other branch under a different summation order and moves the gradient inside that pixel's
        # receptive field (about 50 of 24576 values here).  The fp64 oracle differs from the reference's own fp32
        # golden the same way, in another patch, by up to 1.6e-2 - so the gate is the L2 error plus a bound on how
        # many values may sit outside the north_star tolerance, not the raw maximum.
        assert e2 < 1e-3 and frac < 5e-3 and e < 2e-2
    else:
        assert e2 < B16["dx"]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_aesrgan_gan_steps_fused_trainer(golden_dir, dtype):
    """GanTrainer.step with the attention U-Net discriminator == two iterations of A-ESRGAN/train_aesrgan.py:396-483
    (golden captured from the reference modules; content loss stubbed to 0; aesrgan_config.py:137-155 hyper-parameters)"""
    from sr_gan_fd_amd import model as M
    from sr_gan_fd_amd.gan import GanTrainer
    g = load_golden(golden_dir, "aesrgan_gan_steps.npz")
    f32 = dtype == torch.float32
    torch.manual_seed(0)
    d = M.uNetDiscriminatorAesrgan()
    gen = M.bsrgan_x4(in_channels=3, out_channels=3, channels=64, growth_channels=32, num_rrdb=2)
    scaled_init(gen, 3.0, 0.5)
    d.compute_dtype = gen.compute_dtype = dtype
    gen.cuda().train()
    d.cuda().train()
    tr = GanTrainer(gen, d, None, g_lr=5e-5, d_lr=1e-5, pixel_weight=10.0, adversarial_weight=0.1)
    for it in range(2):
        s = tr.step(torch.tensor(g[f"it{it}_lr"]).cuda(), torch.tensor(g[f"it{it}_gt"]).cuda()).cpu().numpy()
        want = g[f"it{it}_scalars"]  # d_loss, pixel, content, adv, D(gt), D(sr)
        got = [s[0] + s[1], s[2], 0.0, s[3], s[4], s[5]]
        e_sr = _rel(tr.sr, g[f"it{it}_sr"])
        print(f"A-ESRGAN GAN {dtype} it{it}: got {got} want {list(want)} SR err {e_sr:.2e}")
        # float16 (the benchmarked dtype) is held to BASELINE.json's 1e-3 on the logged scalars and SR like f32 (measured 2e-5 / see print);
        # bfloat16 (8 mantissa bits) to 3e-2
        tol16 = 1e-3 if dtype == torch.float16 else 3e-2
        assert np.allclose(got, want, rtol=1e-3 if f32 else tol16, atol=1e-5)
        assert e_sr < (1e-3 if f32 else tol16)
        if not f32:
            continue
        assert _rel(gen.conv4.bias, g[f"it{it}_g_conv4_bias"]) < 1e-3
        assert _rel(d.conv9.weight, g[f"it{it}_d_probe"]) < 1e-3
        # iteration 1 carries the LeakyReLU-mask sensitivity measured in tests/test_oracle_golden.py::test_aesrgan_gan_steps
        at = 2e-4 if it == 0 else 2e-2
        for sd, key in ((gen.state_dict(), f"it{it}_wsum_g"), (d.state_dict(), f"it{it}_wsum_d")):
            for k, want_c in table(g, key).items():
                assert np.allclose(checksum(sd[k]), want_c, rtol=2e-3, atol=at * abs(want_c[1]) + 1e-9), f"{key} {k}: {checksum(sd[k])} vs {want_c}"


@pytest.mark.parametrize("size", [(120, 120), (72, 104)])
def test_aesrgan_discriminator_other_input_sizes(size):
    """aesrgan_config.py:77 trains on 120x120 crops (15x15 at the bottleneck, 17x17 gating map): odd intermediate sizes and a
    non-square input against the CPU oracle"""
    from oracle import srgan_oracle as O
    from sr_gan_fd_amd import model as M
    torch.manual_seed(0)
    d = M.uNetDiscriminatorAesrgan()
    d.compute_dtype = torch.float32
    P = {k: v.detach().clone() for k, v in d.state_dict().items()}
    d.cuda().train()
    x = torch.rand(2, 3, *size)
    xin = x.clone().cuda().requires_grad_(True)
    out = d(xin)
    F.binary_cross_entropy_with_logits(out, torch.ones_like(out)).backward()
    xo = x.clone().requires_grad_(True)
    want = O.aesrgan_unet_forward(xo, P, training=True)
    F.binary_cross_entropy_with_logits(want, torch.ones_like(want)).backward()
    assert out.shape == want.shape == (2, 1, *size)
    assert _rel(out, want) < 1e-4
    assert _rel_l2(xin.grad, xo.grad) < 1e-2          # L2: activation-mask ties (see test_aesrgan_discriminator)


def test_aesrgan_iteration_at_the_config_crop_f16_vs_oracle():
    """aesrgan_config.py:62,102-103: x2, LR crops of 60 x 60 (GT 120 x 120), the 23-RRDB generator with the attention U-Net discriminator.
    One GAN iteration (train_aesrgan.py:396-483, content term off) in float16 -- the scripts' autocast dtype -- against the fp32 CPU
    oracle; batch 2 instead of 8 keeps the oracle at seconds.  At this crop the generator's dense blocks run through the dense-block
    launch (8 x 16 tiles), forward and data gradient, inside the GAN trainer.  Asserted: every logged scalar within 1e-3 relative
    (BASELINE.json's tolerance), probed parameters after both Adam steps within 5e-3 of the tensor's range."""
    from oracle import srgan_oracle as O
    from sr_gan_fd_amd import model as M, ops
    from sr_gan_fd_amd.gan import GanTrainer
    from tests.util import sd_to_params
    torch.manual_seed(9)
    lr_img, gt = torch.rand(2, 3, 60, 60), torch.rand(2, 3, 120, 120)
    torch.manual_seed(0)
    d = M.uNetDiscriminatorAesrgan()
    gen = M.bsrgan_x2(in_channels=3, out_channels=3, channels=64, growth_channels=32, num_rrdb=23)
    scaled_init(gen, 3.0, 0.5)
    G = sd_to_params(gen.state_dict())
    D = {k: v.detach().clone() for k, v in d.state_dict().items()}
    g_opt, d_opt = O.AdamState(G, O.g_param_names(G)), O.AdamState(D, O.d_param_names(D))
    out = O.gan_step(G, D, g_opt, d_opt, lr_img, gt, upscale=2, g_lr=5e-5, d_lr=1e-5, betas=(0.9, 0.999), eps=1e-4, pixel_weight=10.0,
                     content_weight=1.0, adversarial_weight=0.1, d_forward=O.aesrgan_unet_forward)
    d.compute_dtype = gen.compute_dtype = torch.float16
    gen, d = gen.cuda().train(), d.cuda().train()
    tr = GanTrainer(gen, d, None, g_lr=5e-5, d_lr=1e-5, pixel_weight=10.0, adversarial_weight=0.1)
    s = tr.step(lr_img.cuda(), gt.cuda()).cpu().numpy()
    sp = tr.ge._last
    n_chain = len([a for a in sp.fw if type(a) is ops.DenseChain]) + len([it for it in sp.bw if it[0] == "chain"])
    assert n_chain == 6 * 23 or ops.DENSE_CHAIN == "0", n_chain
    got = [s[0] + s[1], s[2], s[3], s[4], s[5]]
    want = [out[k] for k in ("d_loss", "pixel_loss", "adversarial_loss", "d_gt_probability", "d_sr_probability")]
    err = max(abs(a - b) / max(abs(b), 1e-6) for a, b in zip(got, want))
    e_g, e_d = _rel(gen.conv4.bias, G["conv4.bias"]), _rel(d.conv9.weight, D["conv9.weight"])
    print(f"f16 A-ESRGAN iteration at 60 -> 120 x2, 23 RRDB: got {got} want {want} worst rel {err:.2e}; G conv4.bias {e_g:.2e}, D conv9.weight {e_d:.2e}")
    assert err < 1e-3 and e_g < 5e-3 and e_d < 5e-3
