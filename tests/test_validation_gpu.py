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
