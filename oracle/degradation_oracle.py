"""CPU oracle for the on-device degradation stages.  TEST INFRASTRUCTURE ONLY.

A torch-CPU restatement of ``Real_ESRGAN/imgproc.py``'s filter2d_torch / USMSharp / DiffJPEG and the closing
quantisation of degradation_process.  Only ``tests/`` may import it; the product (``sr_gan_fd_amd``) never does.

Pinned by ``tests/golden/degradation.npz`` -- outputs of the reference's own functions, captured by importing
``Real_ESRGAN/imgproc.py`` in the build container (``tests/golden/make_golden.py``; cv2 / torchvision / scipy.stats stubs:
none of the captured functions touches them, except ``USMSharp.__init__``'s ``cv2.getGaussianKernel``, which is
bypassed by registering the kernel OpenCV documents for those arguments).

Every function cites the reference file:line (relative to /root/reference) it follows.
"""
from __future__ import annotations

import itertools

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def filter2d(image: Tensor, kernel: Tensor) -> Tensor:
    """filter2d_torch -- Real_ESRGAN/imgproc.py:1092-1124: reflect pad k//2, one shared kernel or one per image,
    cross-correlation per channel (grouped conv2d)."""
    k = kernel.size(-1)
    b, c, h, w = image.size()
    if k % 2 != 1:
        raise ValueError("Wrong kernel size.")
    image = F.pad(image, (k // 2, k // 2, k // 2, k // 2), mode="reflect")
    ph, pw = image.size()[-2:]
    if kernel.size(0) == 1:
        return F.conv2d(image.view(b * c, 1, ph, pw), kernel.view(1, 1, k, k), padding=0).view(b, c, h, w)
    image = image.view(1, b * c, ph, pw)
    kernel = kernel.view(b, 1, k, k).repeat(1, c, 1, 1).view(b * c, 1, k, k)
    return F.conv2d(image, kernel, groups=b * c).view(b, c, h, w)


def usm_kernel(radius: int = 50, sigma: float = 0) -> Tensor:
    """USMSharp.__init__ -- Real_ESRGAN/imgproc.py:1519-1527: odd radius, cv2.getGaussianKernel(radius, sigma) outer
    product as a (1, r, r) fp32 buffer.  OpenCV (third party, absent) documents the kernel as
    exp(-(i-(r-1)/2)^2 / (2 sigma^2)) normalised, with sigma = 0.3*((r-1)*0.5 - 1) + 0.8 when sigma <= 0."""
    if radius % 2 == 0:
        radius += 1
    if sigma <= 0:
        sigma = 0.3 * ((radius - 1) * 0.5 - 1) + 0.8
    x = np.arange(radius, dtype=np.float64) - (radius - 1) * 0.5
    g = np.exp(-(x * x) / (2.0 * sigma * sigma))
    g = (g / g.sum()).reshape(radius, 1)
    return torch.FloatTensor(np.dot(g, g.transpose())).unsqueeze_(0)


def usm_sharp(x: Tensor, kernel: Tensor, weight: float, threshold: float) -> Tensor:
    """USMSharp.forward -- Real_ESRGAN/imgproc.py:1529-1540"""
    blur = filter2d(x, kernel)
    residual = x - blur
    mask = (torch.abs(residual) * 255 > threshold).float()
    soft_mask = filter2d(mask, kernel)
    out = torch.clip(x + weight * residual, 0, 1)
    return soft_mask * out + (1 - soft_mask) * x


# quantisation tables -- Real_ESRGAN/imgproc.py:43-52 (the standard JPEG tables, transposed as the reference does)
Y_TABLE = torch.from_numpy(np.array(
    [[16, 11, 10, 16, 24, 40, 51, 61], [12, 12, 14, 19, 26, 58, 60, 55], [14, 13, 16, 24, 40, 57, 69, 56],
     [14, 17, 22, 29, 51, 87, 80, 62], [18, 22, 37, 56, 68, 109, 103, 77], [24, 35, 55, 64, 81, 104, 113, 92],
     [49, 64, 78, 87, 103, 121, 120, 101], [72, 92, 95, 98, 112, 100, 103, 99]], dtype=np.float32).T.copy())
_c = np.full((8, 8), 99, dtype=np.float32)
_c[:4, :4] = np.array([[17, 18, 24, 47], [18, 21, 26, 66], [24, 26, 56, 99], [47, 66, 99, 99]]).T
C_TABLE = torch.from_numpy(_c)


def quality_to_factor(quality: Tensor) -> Tensor:
    """_calculate_quality_factor per element -- Real_ESRGAN/imgproc.py:1127-1144 (as DiffJPEG.forward applies it, :1476-1480)"""
    q = quality.clone()
    for i in range(q.size(0)):
        q[i] = (5000. / q[i] if q[i] < 50 else 200. - q[i] * 2) / 100.
    return q


def _dct_tensors():
    fwd = np.zeros((8, 8, 8, 8), dtype=np.float32)
    inv = np.zeros((8, 8, 8, 8), dtype=np.float32)
    for x, y, u, v in itertools.product(range(8), repeat=4):
        fwd[x, y, u, v] = np.cos((2 * x + 1) * u * np.pi / 16) * np.cos((2 * y + 1) * v * np.pi / 16)
        inv[x, y, u, v] = np.cos((2 * u + 1) * x * np.pi / 16) * np.cos((2 * v + 1) * y * np.pi / 16)
    alpha = np.array([1. / np.sqrt(2)] + [1] * 7)
    return (torch.from_numpy(fwd), torch.from_numpy(inv), torch.from_numpy(np.outer(alpha, alpha) * 0.25).float(),
            torch.from_numpy(np.outer(alpha, alpha)).float())


def diff_jpeg(x: Tensor, factor: Tensor, differentiable: bool = False) -> Tensor:
    """DiffJPEG.forward after the quality -> factor step -- Real_ESRGAN/imgproc.py:1482-1497 with _CompressJPEG
    (:1300-1324) and _DeCompressJPEG (:1427-1462).  x (B,3,H,W) in [0,1]; factor (B,)."""
    fwd, inv, scale, alpha = _dct_tensors()
    b = x.size(0)
    h, w = x.size()[-2:]
    h_pad = (16 - h % 16) % 16
    w_pad = (16 - w % 16) % 16
    x = F.pad(x, (0, w_pad, 0, h_pad), mode="constant", value=0)
    hh, ww = h + h_pad, w + w_pad
    # :1201-1212 RGB -> YCbCr on the x255 image, channels last
    m = torch.from_numpy(np.array([[0.299, 0.587, 0.114], [-0.168736, -0.331264, 0.5], [0.5, -0.418688, -0.081312]], dtype=np.float32).T.copy())
    ycc = torch.tensordot((x * 255).permute(0, 2, 3, 1), m, dims=1) + torch.tensor([0., 128., 128.])
    # :1219-1226 chroma average
    img = ycc.permute(0, 3, 1, 2)
    comps = {"y": ycc[:, :, :, 0], "cb": F.avg_pool2d(img[:, 1:2], 2, 2).squeeze(1), "cr": F.avg_pool2d(img[:, 2:3], 2, 2).squeeze(1)}

    def rnd(t):
        r = torch.round(t)
        return r + (t - r) ** 3 if differentiable else r           # :1183-1195
    out = {}
    for k, comp in comps.items():
        ch, cw = comp.shape[1:3]
        blocks = comp.view(b, ch // 8, 8, -1, 8).permute(0, 1, 3, 2, 4).contiguous().view(b, -1, 8, 8)        # :1235-1242
        coef = scale * torch.tensordot(blocks - 128, fwd, dims=2)                                              # :1254-1259
        table = (Y_TABLE if k == "y" else C_TABLE).expand(b, 1, 8, 8) * factor.view(b, 1, 1, 1)                # :1270-1278
        q = rnd(coef.float() / table)
        deq = q * table                                                                                        # :1333-1340
        pix = 0.25 * torch.tensordot(deq * alpha, inv, dims=2) + 128                                           # :1370-1374
        out[k] = pix.view(b, ch // 8, cw // 8, 8, 8).permute(0, 1, 3, 2, 4).contiguous().view(b, ch, cw)       # :1382-1388

    def rep(t):
        return t.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)                                       # :1396-1405
    ycc2 = torch.stack([out["y"], rep(out["cb"]), rep(out["cr"])], dim=3)
    m2 = torch.from_numpy(np.array([[1., 0., 1.402], [1, -0.344136, -0.714136], [1, 1.772, 0]], dtype=np.float32).T.copy())
    rgb = torch.tensordot(ycc2 + torch.tensor([0, -128., -128.]), m2, dims=1).permute(0, 3, 1, 2)              # :1414-1424
    rgb = torch.clamp(rgb, 0, 255) / 255                                                                       # :1459-1460
    return rgb[:, :, 0:h, 0:w].contiguous()   # (the reference returns this permuted view; its CPU filter2d_torch cannot .view() it)


def quantize_u8(x: Tensor) -> Tensor:
    """last line of degradation_process -- Real_ESRGAN/imgproc.py:2460"""
    return torch.clamp((x * 255.0).round(), 0, 255) / 255.


def interpolate(x: Tensor, size=None, scale_factor=None, mode: str = "bilinear") -> Tensor:
    """the F_torch.interpolate calls of degradation_process -- Real_ESRGAN/imgproc.py:2374, :2415-2418, :2440-2442, :2454-2456
    (torch's own CPU kernels are the statement of "area" / "bilinear" / "bicubic" here)"""
    return F.interpolate(x, size=size, scale_factor=scale_factor, mode=mode)


def _finish(out: Tensor, clip: bool, rounds: bool) -> Tensor:
    if clip and rounds:
        return torch.clamp((out * 255.0).round(), 0, 255) / 255.
    if clip:
        return torch.clamp(out, 0, 1)
    if rounds:
        return (out * 255.0).round() / 255.
    return out


def add_gaussian_noise(image: Tensor, sigma, clip: bool = True, rounds: bool = False, gray_noise=0) -> Tensor:
    """_add_gaussian_noise_torch over _generate_gaussian_noise_torch -- Real_ESRGAN/imgproc.py:970-998, :832-866.  Draws:
    randn(h, w) first when any image wants grey noise, then randn(b, c, h, w)."""
    b, _, h, w = image.size()
    if not isinstance(sigma, (float, int)):
        sigma = sigma.view(b, 1, 1, 1)
    if isinstance(gray_noise, (float, int)):
        cal_gray = gray_noise > 0
    else:
        gray_noise = gray_noise.view(b, 1, 1, 1)
        cal_gray = torch.sum(gray_noise) > 0
    if cal_gray:
        noise_gray = (torch.randn(h, w, dtype=image.dtype) * sigma / 255.).view(b, 1, h, w)
    noise = torch.randn(*image.size(), dtype=image.dtype) * sigma / 255.
    if cal_gray:
        noise = noise * (1 - gray_noise) + noise_gray * gray_noise
    return _finish(image + noise, clip, rounds)


def rgb_to_grayscale(img: Tensor) -> Tensor:
    """torchvision.transforms.functional_tensor.rgb_to_grayscale (third party, absent; any 0.13-0.16): 0.2989 R + 0.587 G +
    0.114 B, one output channel.  PARITY UNPINNED for the grey Poisson branch that uses it."""
    r, g, b = img.unbind(dim=-3)
    return (0.2989 * r + 0.587 * g + 0.114 * b).to(img.dtype).unsqueeze(dim=-3)


def poisson_vals(image_q: Tensor) -> Tensor:
    """vals = 2 ** ceil(log2(#unique values)) per image -- Real_ESRGAN/imgproc.py:898-901 / :907-910"""
    b = image_q.size(0)
    vals_list = [len(torch.unique(image_q[i])) for i in range(b)]
    return image_q.new_tensor([2 ** np.ceil(np.log2(v)) for v in vals_list]).view(b, 1, 1, 1)


def add_poisson_noise(image: Tensor, scale, clip: bool = True, rounds: bool = False, gray_noise=0) -> Tensor:
    """_add_poisson_noise_torch over _generate_poisson_noise_torch -- Real_ESRGAN/imgproc.py:1001-1029, :869-919"""
    b, _, h, w = image.size()
    if isinstance(gray_noise, (float, int)):
        cal_gray = gray_noise > 0
    else:
        gray_noise = gray_noise.view(b, 1, 1, 1)
        cal_gray = torch.sum(gray_noise) > 0
    if cal_gray:
        img_gray = torch.clamp((rgb_to_grayscale(image) * 255.0).round(), 0, 255) / 255.
        vals = poisson_vals(img_gray)
        noise_gray = (torch.poisson(img_gray * vals) / vals - img_gray).expand(b, 3, h, w)
    image_q = torch.clamp((image * 255.0).round(), 0, 255) / 255.
    vals = poisson_vals(image_q)
    noise = torch.poisson(image_q * vals) / vals - image_q
    if cal_gray:
        noise = noise * (1 - gray_noise) + noise_gray * gray_noise
    if not isinstance(scale, (float, int)):
        scale = scale.view(b, 1, 1, 1)
    # the rounded copy is local to the generator (:904); the caller adds the noise to the image it was given (:1020)
    return _finish(image + noise * scale, clip, rounds)


def random_add_gaussian_noise(image, sigma_range, gray_prob, clip=True, rounds=False):
    """random_add_gaussian_noise_torch -- Real_ESRGAN/imgproc.py:1032-1060, draws at :937-941"""
    b = image.size(0)
    sigma = torch.rand(b, dtype=image.dtype) * (sigma_range[1] - sigma_range[0]) + sigma_range[0]
    gray = (torch.rand(b, dtype=image.dtype) < gray_prob).float()
    return add_gaussian_noise(image, sigma, clip, rounds, gray)


def random_add_poisson_noise(image, scale_range, gray_prob, clip=True, rounds=False):
    """random_add_poisson_noise_torch -- Real_ESRGAN/imgproc.py:1063-1089, draws at :961-965"""
    b = image.size(0)
    scale = torch.rand(b, dtype=image.dtype) * (scale_range[1] - scale_range[0]) + scale_range[0]
    gray = (torch.rand(b, dtype=image.dtype) < gray_prob).float()
    return add_poisson_noise(image, scale, clip, rounds, gray)


def degradation_process(gt, gaussian_kernel1, gaussian_kernel2, sinc_kernel, upscale_factor, P, usm=None):
    """degradation_process -- Real_ESRGAN/imgproc.py:2323-2462 (jpeg_operation = DiffJPEG(False)); `usm` = (kernel, weight,
    threshold) or None.  Host draws (np.random / random) in the reference's order."""
    import random
    H, W = gt.size()[2:4]
    gt_usm = usm_sharp(gt, *usm) if usm is not None else gt
    out = gt_usm
    if np.random.uniform() <= P["first_blur_probability"]:
        out = filter2d(gt_usm, gaussian_kernel1)

    def pick_scale(prob, rng):
        t = random.choices(["up", "down", "keep"], prob)[0]
        return np.random.uniform(1, rng[1]) if t == "up" else np.random.uniform(rng[0], 1) if t == "down" else 1

    def noise(x, i):
        if np.random.uniform() < P[f"gaussian_noise_probability{i}"]:
            return random_add_gaussian_noise(x, P[f"noise_range{i}"], P[f"gray_noise_probability{i}"])
        return random_add_poisson_noise(x, P[f"poisson_scale_range{i}"], P[f"gray_noise_probability{i}"])

    def jpeg(x, rng):
        q = x.new_zeros(x.size(0)).uniform_(*rng)
        return diff_jpeg(torch.clamp(x, 0, 1), quality_to_factor(q))
    scale = pick_scale(P["resize_probability1"], P["resize_range1"])
    out = interpolate(out, scale_factor=scale, mode=random.choice(["area", "bilinear", "bicubic"]))
    out = jpeg(noise(out, 1), P["jpeg_range1"])
    if np.random.uniform() < P["second_blur_probability"]:
        out = filter2d(out, gaussian_kernel2)
    scale = pick_scale(P["resize_probability2"], P["resize_range2"])
    out = interpolate(out, size=(int(H / upscale_factor * scale), int(W / upscale_factor * scale)), mode=random.choice(["area", "bilinear", "bicubic"]))
    out = noise(out, 2)
    if np.random.uniform() < 0.5:
        out = interpolate(out, size=(H // upscale_factor, W // upscale_factor), mode=random.choice(["area", "bilinear", "bicubic"]))
        out = jpeg(filter2d(out, sinc_kernel), P["jpeg_range2"])
    else:
        out = jpeg(out, P["jpeg_range2"])
        out = interpolate(out, size=(H // upscale_factor, W // upscale_factor), mode=random.choice(["area", "bilinear", "bicubic"]))
        out = filter2d(out, sinc_kernel)
    return gt_usm, gt, quantize_u8(out)


def random_crop_lists(gts, lrs, gt_patch_size: int, upscale_factor: int):
    """random_crop_torch on tensor lists -- Real_ESRGAN/imgproc.py:2081-2155: LR window drawn with random.randint (row, column)"""
    import random
    lh, lw = lrs[0].size()[-2:]
    lps = gt_patch_size // upscale_factor
    top, left = random.randint(0, lh - lps), random.randint(0, lw - lps)
    gtop, gleft = int(top * upscale_factor), int(left * upscale_factor)
    return [v[:, :, gtop:gtop + gt_patch_size, gleft:gleft + gt_patch_size] for v in gts], [v[:, :, top:top + lps, left:left + lps] for v in lrs]


def rotate_flip(t: Tensor, op: int) -> Tensor:
    """random_rotate_torch / random_*_flip_torch on square even-sized tensors -- Real_ESRGAN/imgproc.py:2158-2320.  The reference
    calls torchvision (third party, absent): F.rotate(img, angle, center=[w // 2, h // 2]) rotates counter-clockwise about what is
    the exact image centre in torchvision's half-pixel convention, i.e. torch.rot90 for multiples of 90 degrees; hflip / vflip
    reverse the last / second-to-last axis.  PARITY UNPINNED (restated from torchvision's documentation).
    op 1 / 2 / 3 = 90 / 180 / 270 degrees, 4 = hflip, 5 = vflip."""
    if op in (1, 2, 3):
        return torch.rot90(t, op, dims=(-2, -1))
    return torch.flip(t, dims=(-1,)) if op == 4 else torch.flip(t, dims=(-2,)) if op == 5 else t
