"""GPU parity of the validation / data-side pieces: PSNR, random_crop, checkpoint loading and
forward-only inference at sizes that are not multiples of the tile."""
import os
import random

import numpy as np
import pytest
import torch

from tests.util import load_golden, scaled_init

pytestmark = pytest.mark.gpu


def test_psnr_matches_reference(golden_dir):
    from sr_gan_fd_amd.image_quality_assessment import PSNR
    g = load_golden(golden_dir, "validation.npz")
    a, b = torch.tensor(g["psnr_a"]).cuda(), torch.tensor(g["psnr_b"]).cuda()
    for cb, y, key in ((4, True, "psnr_y_cb4"), (4, False, "psnr_rgb_cb4"), (0, True, "psnr_y_cb0")):
        got = PSNR(cb, y)(a, b).cpu().numpy()
        print(key, got, g[key])
        assert got.dtype == np.float64 and np.allclose(got, g[key], rtol=0, atol=1e-4)   # dB; the luma is fp32 in both


def test_ssim_matches_reference(golden_dir):
    """SSIM module vs values captured from the reference's _ssim_single_torch (fp64 maths on both sides; the result is fp32)"""
    from oracle import srgan_oracle as O
    from sr_gan_fd_amd.image_quality_assessment import SSIM, gaussian_kernel_1d
    g = load_golden(golden_dir, "validation.npz")
    a, b = torch.tensor(g["psnr_a"]).cuda(), torch.tensor(g["psnr_b"]).cuda()
    assert np.allclose(SSIM(0, True).gaussian_kernel_window, g["ssim_window"], rtol=0, atol=1e-16)
    assert gaussian_kernel_1d(11, 1.5).shape == (11, 1)
    for cb, y, key in ((4, True, "ssim_y_cb4"), (4, False, "ssim_rgb_cb4"), (0, True, "ssim_y_cb0")):
        got = SSIM(cb, y)(a, b).cpu().numpy()
        print(key, got, g[key])
        assert got.dtype == np.float32 and np.allclose(got, g[key], rtol=0, atol=1e-6), key
    got = SSIM(0, True)(a, torch.roll(b, 3, dims=3)).cpu().numpy()
    assert np.allclose(got, g["ssim_y_rolled"], rtol=0, atol=1e-6)
    box = SSIM(2, False, window_size=7)
    box.gaussian_kernel_window = np.full((7, 7), 1.0 / 49.0)            # any 2-D window, as _ssim_torch accepts
    assert np.allclose(box(a, b).cpu().numpy(), g["ssim_box7_rgb_cb2"], rtol=0, atol=1e-6)
    assert np.allclose(SSIM(0, True)(a, a).cpu().numpy(), 1.0, rtol=0, atol=1e-6)
    # ragged sizes (map not a multiple of the 16x16 tile, one-pixel map) against the oracle
    torch.manual_seed(3)
    for shape, cb in (((2, 3, 11, 11), 0), ((1, 3, 29, 75), 3), ((3, 3, 64, 33), 0)):
        x = torch.rand(shape)
        y2 = (x + 0.1 * torch.randn(shape)).clamp(0, 1)
        for yo in (True, False):
            want = O.ssim(x, y2, cb, yo).numpy()
            got = SSIM(cb, yo)(x.cuda(), y2.cuda()).cpu().numpy()
            assert np.allclose(got, want, rtol=0, atol=1e-6), (shape, cb, yo, got, want)
    with pytest.raises(Exception):
        SSIM(0, True)(torch.rand(1, 3, 8, 8).cuda(), torch.rand(1, 3, 8, 8).cuda())     # 11x11 window does not fit


def test_random_crop_matches_reference(golden_dir):
    from sr_gan_fd_amd import imgproc
    g = load_golden(golden_dir, "validation.npz")
    gt, lr = torch.tensor(g["crop_gt"]).cuda(), torch.tensor(g["crop_lr"]).cuda()
    for seed in (7, 11):
        random.seed(seed)
        pg, pl = imgproc.random_crop(gt, lr, 32, 4)
        assert np.array_equal(pg.cpu().numpy(), g[f"crop{seed}_gt"]) and np.array_equal(pl.cpu().numpy(), g[f"crop{seed}_lr"])
    random.seed(3)
    pg, pl = imgproc.random_crop(gt[:, :, :32, :32].contiguous(), lr[:, :, :8, :8].contiguous(), 32, 4)   # identity-sized window
    assert torch.equal(pg, gt[:, :, :32, :32]) and torch.equal(pl, lr[:, :, :8, :8])


def test_checkpoint_roundtrip_and_ragged_inference(tmp_path, golden_dir):
    """utils.load_state_dict semantics (shape-mismatched / unknown keys dropped) + eval forward at 1x3x37x52 vs the oracle"""
    from oracle import srgan_oracle as O
    from sr_gan_fd_amd import model as M
    from sr_gan_fd_amd.utils import load_state_dict
    from sr_gan_fd_amd.image_quality_assessment import PSNR
    torch.manual_seed(0)
    src = M.bsrgan_x4(in_channels=3, out_channels=3, channels=64, growth_channels=32, num_rrdb=2)
    scaled_init(src, 3.0, 0.5)
    sd = {k: v.clone() for k, v in src.state_dict().items()}
    sd["not.a.key"] = torch.zeros(3)
    sd["conv1.bias"] = torch.zeros(7)                       # wrong shape: must be ignored, not raise
    path = os.path.join(tmp_path, "g.pth.tar")
    torch.save({"state_dict": sd, "epoch": 3}, path)
    torch.manual_seed(1)
    dst = M.bsrgan_x4(in_channels=3, out_channels=3, channels=64, growth_channels=32, num_rrdb=2)
    bias_before = dst.conv1.bias.detach().clone()
    dst.compute_dtype = torch.float32
    dst = load_state_dict(dst, path).cuda().eval()
    assert torch.equal(dst.conv1.bias.cpu(), bias_before)
    assert torch.equal(dst.conv4.weight.cpu(), src.conv4.weight)
    x = torch.rand(1, 3, 37, 52)
    with torch.no_grad():
        sr = dst(x.cuda())
    P = {k: v.detach().clone() for k, v in dst.state_dict().items()}
    P = {k: v.cpu() for k, v in P.items()}
    want = O.rrdbnet_forward(x, P, 4)
    assert sr.shape == (1, 3, 148, 208)
    err = (sr.cpu() - want).abs().max().item()
    print("ragged inference max err", err, "PSNR vs oracle", PSNR(4, True)(sr, want.cuda()).item())
    assert err < 1e-3


def test_parameter_changes_behind_the_engine_are_picked_up():
    """load_state_dict after a forward, in-place edits, and a .cpu()/.cuda() round trip must all reach the packed weights"""
    from sr_gan_fd_amd import model as M
    torch.manual_seed(0)
    a = M.bsrgan_x4(num_rrdb=1)
    scaled_init(a, 3.0, 0.5)
    torch.manual_seed(1)
    b = M.bsrgan_x4(num_rrdb=1)
    scaled_init(b, 2.0, 0.4)
    a.compute_dtype = b.compute_dtype = torch.float32
    a.cuda().eval()
    b.cuda().eval()
    x = torch.rand(1, 3, 24, 24, device="cuda")
    with torch.no_grad():
        ya, yb = a(x), b(x)
        assert not torch.equal(ya, yb)
        a.load_state_dict(b.state_dict())                    # engine of `a` already built and packed
        assert torch.equal(a(x), yb)
        a.conv4.bias.add_(0.125)                             # in-place edit
        y2 = a(x)
        assert torch.allclose(y2, (yb + 0.125).clamp(0, 1), atol=1e-6) and not torch.equal(y2, yb)
        a.cpu()
        a.cuda()                                             # parameters moved: the flat buffer is rebuilt
        assert torch.equal(a(x), y2)


def test_cuda_prefetcher_feeds_the_hip_path():
    """dataset.CUDAPrefetcher (interface of BSRGAN/dataset.py:203-243): batches staged on a copy stream are consumed by kernels
    launched through the C ABI on the current stream; values must be those of the source batches, in order, for two epochs"""
    from sr_gan_fd_amd.dataset import CUDAPrefetcher
    from sr_gan_fd_amd.image_quality_assessment import PSNR
    torch.manual_seed(0)
    batches = [{"gt": torch.rand(2, 3, 64, 64).pin_memory(), "lr": torch.rand(2, 3, 16, 16).pin_memory(), "name": f"b{i}"} for i in range(4)]
    pf = CUDAPrefetcher(batches, torch.device("cuda", 0))
    assert len(pf) == 4
    psnr = PSNR(0, True)
    for epoch in range(2):
        seen = 0
        batch = pf.next()
        while batch is not None:
            src = batches[seen]
            assert batch["name"] == src["name"] and batch["gt"].is_cuda and batch["lr"].is_cuda
            # identical images -> the PSNR kernel's error term is the 1e-8 floor: 10*log10(255^2 / 1e-8)
            val = psnr(batch["gt"], src["gt"].cuda())
            assert torch.allclose(val, torch.full_like(val, 10 * np.log10(255.0 ** 2 / 1e-8)), atol=1e-6)
            assert torch.equal(batch["lr"].cpu(), src["lr"])
            seen += 1
            batch = pf.next()
        assert seen == 4
        pf.reset()


def test_u8_ingest_equals_the_references_host_path():
    """imgproc.image_to_tensor_u8 == the reference's per-image host ingest (dataset.py:66,81,90 + imgproc.py:331-358): uint8 HWC BGR ->
    astype(float32) / 255 -> crop -> BGR2RGB -> image_to_tensor(range_norm, half=False), restated with numpy / torch on the CPU; bitwise,
    at a ragged window of odd-sized images, with and without the [-1, 1] range; the prefetcher's ingest_u8 mode hands out the same batch."""
    from sr_gan_fd_amd.dataset import CUDAPrefetcher
    from sr_gan_fd_amd.imgproc import image_to_tensor_u8
    rng = np.random.default_rng(0)
    imgs = rng.integers(0, 256, size=(3, 37, 53, 3), dtype=np.uint8)
    top, left, ph, pw = 5, 9, 24, 40

    def host(image_u8, range_norm):
        im = image_u8.astype(np.float32) / 255.                       # dataset.py:66
        im = im[top:top + ph, left:left + pw, ...]                    # imgproc.random_crop_np's slicing
        im = im[..., ::-1]                                            # cv2.COLOR_BGR2RGB
        t = torch.from_numpy(np.ascontiguousarray(im)).permute(2, 0, 1).float()      # imgproc.py:348
        return t.mul(2.0).sub(1.0) if range_norm else t               # :351-352
    dev = torch.tensor(imgs).cuda()
    for rn in (False, True):
        want = torch.stack([host(im, rn) for im in imgs])
        got = image_to_tensor_u8(dev, top, left, (ph, pw), bgr=True, range_norm=rn)
        assert got.shape == (3, 3, ph, pw) and torch.equal(got.cpu(), want)
    full = image_to_tensor_u8(dev, bgr=False)
    assert torch.equal(full.cpu(), torch.tensor(imgs).permute(0, 3, 1, 2).float() / 255.)
    batches = [{"gt": torch.tensor(imgs).pin_memory(), "name": "b0"}]
    b = CUDAPrefetcher(batches, torch.device("cuda", 0), ingest_u8=True).next()
    assert b["gt"].dtype == torch.float32 and torch.equal(b["gt"].cpu(), torch.tensor(np.ascontiguousarray(imgs[..., ::-1])).permute(0, 3, 1, 2).float() / 255.)
