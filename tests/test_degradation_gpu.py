"""GPU parity of the on-device degradation stages: filter2d_torch, USMSharp, DiffJPEG against outputs of the
reference's own functions (tests/golden/degradation.npz) and against the CPU oracle at other sizes.  fp32 everywhere;
tolerance 1e-5 absolute on [0,1] images (summation order), with a bounded fraction of pixels allowed to differ where the
algorithm itself is discontinuous (the USM threshold mask, JPEG's rounding) -- stated per test."""
import numpy as np
import pytest
import torch

from tests.util import load_golden

pytestmark = pytest.mark.gpu


def _cmp(got, want, atol, what, max_bad_frac=0.0):
    got, want = got.detach().cpu().double().numpy(), np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = np.abs(got - want)
    bad = float((err > atol).mean())
    print(f"{what}: max err {err.max():.2e}, fraction above {atol:g}: {bad:.2e}")
    assert bad <= max_bad_frac, (what, err.max(), bad)


def test_filter2d_matches_reference(golden_dir):
    from sr_gan_fd_amd import imgproc
    g = load_golden(golden_dir, "degradation.npz")
    img = torch.tensor(g["image"]).cuda()
    k21 = torch.tensor(g["kernels21"]).cuda()
    _cmp(imgproc.filter2d_torch(img, k21), g["filter2d_per_image"], 1e-5, "per-image 21x21")
    _cmp(imgproc.filter2d_torch(img, k21[:1]), g["filter2d_shared"], 1e-5, "shared 21x21")
    _cmp(imgproc.filter2d_torch(img, torch.tensor(g["kernel7"]).cuda()), g["filter2d_k7"], 2e-5, "shared 7x7 (unnormalised, sums to ~25)")
    with pytest.raises(ValueError, match="Wrong kernel size."):
        imgproc.filter2d_torch(img, torch.rand(1, 4, 4).cuda())
    with pytest.raises(Exception):
        imgproc.filter2d_torch(img[:, :, :8, :8].contiguous(), k21)      # reflect padding of 10 needs more than 8 pixels


def test_filter2d_other_sizes_vs_oracle():
    """tile edges (32x64 output tiles), the largest kernel, one-channel images, identity and shift kernels"""
    from oracle import degradation_oracle as D
    from sr_gan_fd_amd import imgproc
    torch.manual_seed(5)
    for shape, k, per_image in (((2, 3, 32, 64), 3, True), ((1, 1, 33, 65), 5, False), ((4, 3, 100, 130), 21, True), ((2, 2, 64, 60), 51, False),
                                ((1, 3, 27, 26), 51, True)):
        img = torch.rand(shape)
        ker = torch.rand(shape[0] if per_image else 1, k, k)
        ker = ker / ker.sum(dim=(1, 2), keepdim=True)
        _cmp(imgproc.filter2d_torch(img.cuda(), ker.cuda()), D.filter2d(img, ker).numpy(), 1e-5, f"{shape} k={k}")
    img = torch.rand(2, 3, 40, 70)
    ident = torch.zeros(1, 9, 9)
    ident[0, 4, 4] = 1
    assert torch.equal(imgproc.filter2d_torch(img.cuda(), ident.cuda()).cpu(), img)
    shift = torch.zeros(1, 9, 9)
    shift[0, 4, 0] = 1                                                     # out[y, x] = in[y, reflect(x - 4)]
    got = imgproc.filter2d_torch(img.cuda(), shift.cuda()).cpu()
    assert torch.equal(got[..., 4:], img[..., :-4]) and torch.equal(got[..., 0], img[..., 4]) and torch.equal(got[..., 3], img[..., 1])


def test_usm_sharp_matches_reference(golden_dir):
    """The |residual|*255 > threshold mask is a step function: a residual within 1e-6 of the threshold may fall on the other
    side, and the 51x51 blur of the mask spreads that over its window with weight <= 1/400 -- bounded at 1e-4 here."""
    from sr_gan_fd_amd import imgproc
    g = load_golden(golden_dir, "degradation.npz")
    usm = imgproc.USMSharp().cuda()
    assert usm.radius == 51 and tuple(usm.kernel.shape) == (1, 51, 51)
    _cmp(usm.kernel, g["usm_kernel"], 1e-9, "USM kernel (OpenCV's documented Gaussian)")
    big = torch.tensor(g["usm_image"]).cuda()
    _cmp(usm(big, 0.5, 10), g["usm_w05_t10"], 1e-4, "usm w=0.5 t=10")
    _cmp(usm(big, 1.5, 3), g["usm_w15_t3"], 1e-4, "usm w=1.5 t=3")
    _cmp(usm(big, 0.5, 10), g["usm_w05_t10"], 1e-5, "usm w=0.5 t=10 (tight, few pixels)", max_bad_frac=0.02)
    flat = torch.full((1, 3, 64, 64), 0.25).cuda()
    assert torch.allclose(usm(flat, 0.5, 10), flat, atol=1e-6)              # nothing to sharpen


def _jpeg_check(got, want, what):
    """A coefficient whose quotient lands within float rounding of .5 may round the other way (torch.round on a 64-term fp32
    sum): that moves one 8x8 block (a 16x16 area for chroma) by about one quantisation step.  Gate: 99.5 % of the pixels
    within 1e-5, and no pixel further than one coarse step (0.1)."""
    _cmp(got, want, 1e-5, what, max_bad_frac=5e-3)
    assert float((got.detach().cpu() - torch.tensor(want)).abs().max()) < 0.1


def test_diff_jpeg_matches_reference(golden_dir):
    from sr_gan_fd_amd import imgproc
    g = load_golden(golden_dir, "degradation.npz")
    img = torch.tensor(g["image"]).cuda()
    jp = imgproc.DiffJPEG().cuda()
    q = torch.tensor(g["jpeg_quality"]).cuda()
    out = jp(img, q)
    _jpeg_check(out, g["jpeg"], "jpeg 45x70 q=30/75/95")
    assert np.allclose(q.cpu().numpy(), g["jpeg_factor"], rtol=1e-6)        # quality -> factor in place, like the reference
    _jpeg_check(imgproc.DiffJPEG(True).cuda()(img, torch.tensor(g["jpeg_quality"]).cuda()), g["jpeg_diff"], "jpeg differentiable rounding")
    _jpeg_check(jp(img[:, :, :32, :48].contiguous(), 60), g["jpeg_scalar_q60"], "jpeg scalar quality 60")
    with pytest.raises(Exception):
        jp(img[:, :2].contiguous(), 60)                                     # RGB only


def test_diff_jpeg_other_sizes_vs_oracle():
    from oracle import degradation_oracle as D
    from sr_gan_fd_amd import imgproc
    jp = imgproc.DiffJPEG().cuda()
    torch.manual_seed(9)
    for shape in ((1, 3, 16, 16), (2, 3, 17, 15), (5, 3, 64, 96), (1, 3, 130, 33)):
        img = torch.nn.functional.interpolate(torch.rand(shape[0], 3, 6, 7), size=shape[2:], mode="bilinear") * 0.8 + 0.2 * torch.rand(shape)
        quality = torch.empty(shape[0]).uniform_(20, 98)
        want = D.diff_jpeg(img, D.quality_to_factor(quality))
        _jpeg_check(jp(img.cuda(), quality.cuda()), want.numpy(), f"jpeg {shape}")
    # a flat grey image survives exactly: only the DC terms are non-zero and they quantise back to themselves within a step
    grey = torch.full((1, 3, 32, 32), 0.5).cuda()
    assert float((jp(grey, 90) - grey).abs().max()) < 2e-2
