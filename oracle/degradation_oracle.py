"""CPU oracle for the on-device degradation stages.  TEST INFRASTRUCTURE ONLY.

A torch-CPU restatement of ``Real_ESRGAN/imgproc.py``'s filter2d_torch / USMSharp / DiffJPEG and the closing
quantisation of degradation_process.  Only ``tests/`` may import it; the product (``sr-gan-fd_amd``) never does.

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
    return rgb[:, :, 0:h, 0:w]


def quantize_u8(x: Tensor) -> Tensor:
    """last line of degradation_process -- Real_ESRGAN/imgproc.py:2460"""
    return torch.clamp((x * 255.0).round(), 0, 255) / 255.
