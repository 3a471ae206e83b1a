"""Device-side mirror of the data-side helpers the train loops call on GPU tensors (reference: BSRGAN/imgproc.py for
``random_crop``; Real_ESRGAN/imgproc.py for the on-device degradation stages -- SURVEY 8f N4: ``filter2d_torch``,
``USMSharp``, ``DiffJPEG``).  The CPU-side pieces of that file (kernel synthesis with numpy / scipy, cv2 image I/O)
stay the reference's own."""
from __future__ import annotations

import random
from typing import Tuple, Union

import numpy as np
import torch
from torch import Tensor, nn

from . import _abi as A


def random_crop(gt_tensor: Tensor, lr_tensor: Tensor, gt_image_size: int, upscale_factor: int) -> Tuple[Tensor, Tensor]:
    """imgproc.random_crop (BSRGAN/imgproc.py:846-886): one (top, left) for the whole batch drawn from Python's
    ``random`` stream (row first, then column -- seed per rank under data parallelism), LR window at the
    integer-divided position, outputs in ``lr_tensor.dtype``.  The reference copies B slices in a Python loop; here
    each tensor is one strided-copy launch.  Identity-sized requests return the inputs' data unchanged."""
    h, w = gt_tensor.shape[2], gt_tensor.shape[3]
    top = random.randint(0, h - gt_image_size)
    left = random.randint(0, w - gt_image_size)
    lr_top, lr_left, lr_size = top // upscale_factor, left // upscale_factor, gt_image_size // upscale_factor
    if not (gt_tensor.is_cuda and lr_tensor.is_cuda):
        raise A.SrganfdError("random_crop: tensors must be on the GPU (the HIP library is the product; no CPU fallback)")
    L, st = A.lib(), A.stream_ptr()
    out = []
    for src, t, l, s in ((gt_tensor, top, left, gt_image_size), (lr_tensor, lr_top, lr_left, lr_size)):
        x = src.contiguous().float()
        dst = torch.empty(x.shape[0], x.shape[1], s, s, dtype=torch.float32, device=x.device)
        A.check(L.srganfd_crop_nchw(x.data_ptr(), dst.data_ptr(), x.shape[0], x.shape[1], x.shape[2], x.shape[3], t, l, s, s, st), "crop_nchw")
        out.append(dst.to(lr_tensor.dtype))
    return out[0], out[1]


def _need_gpu(t: Tensor, what: str) -> None:
    if not t.is_cuda:
        raise A.SrganfdError(f"{what}: tensors must be on the GPU (the HIP library is the product; no CPU fallback)")


def filter2d_torch(image: Tensor, kernel: Tensor) -> Tensor:
    """imgproc.filter2d_torch (Real_ESRGAN/imgproc.py:1092-1124): reflect padding + per-image (or shared) k x k
    cross-correlation of every channel.  One LDS-tiled HIP launch instead of pad + view + grouped conv2d; an even
    kernel size raises ``ValueError("Wrong kernel size.")`` like the reference."""
    k = kernel.size(-1)
    b, c, h, w = image.size()
    if k % 2 != 1:
        raise ValueError("Wrong kernel size.")
    _need_gpu(image, "filter2d_torch")
    x = image.detach().contiguous().float()
    kk = kernel.detach().to(device=x.device, dtype=torch.float32).contiguous()
    out = torch.empty_like(x)
    A.check(A.lib().srganfd_filter2d(x.data_ptr(), kk.data_ptr(), 1 if kk.size(0) == 1 else kk.size(0), b, c, h, w, k, out.data_ptr(),
                                     A.stream_ptr()), "filter2d")
    return out


def _gaussian_kernel_1d(ksize: int, sigma: float) -> np.ndarray:
    """what cv2.getGaussianKernel(ksize, sigma) returns per OpenCV's documentation (OpenCV is not a dependency here)"""
    if sigma <= 0:
        sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) * 0.5
    g = np.exp(-(x * x) / (2.0 * sigma * sigma))
    return (g / g.sum()).reshape(ksize, 1)


class USMSharp(nn.Module):
    """imgproc.USMSharp (Real_ESRGAN/imgproc.py:1517-1540): same constructor, ``kernel`` buffer (1, r, r) and
    ``forward(x, weight, threshold)``.  Two fused HIP passes: blur -> residual + threshold mask, then blurred mask ->
    blend (the reference runs two grouped convs and six elementwise ops)."""

    def __init__(self, radius: int = 50, sigma: int = 0) -> None:
        super().__init__()
        if radius % 2 == 0:
            radius += 1
        self.radius = radius
        kernel = _gaussian_kernel_1d(radius, sigma)
        kernel = torch.FloatTensor(np.dot(kernel, kernel.transpose())).unsqueeze_(0)
        self.register_buffer("kernel", kernel)

    def forward(self, x: Tensor, weight: float, threshold: int) -> Tensor:
        _need_gpu(x, "USMSharp")
        xx = x.detach().contiguous().float()
        b, c, h, w = xx.shape
        kk = self.kernel.to(device=xx.device, dtype=torch.float32).contiguous()
        out = torch.empty_like(xx)
        ws = torch.empty(2 * xx.numel(), dtype=torch.float32, device=xx.device)
        A.check(A.lib().srganfd_usm_sharp(xx.data_ptr(), kk.data_ptr(), b, c, h, w, self.radius, float(weight), float(threshold), out.data_ptr(),
                                          ws.data_ptr(), A.stream_ptr()), "usm_sharp")
        return out


class DiffJPEG(nn.Module):
    """imgproc.DiffJPEG (Real_ESRGAN/imgproc.py:1465-1497): ``forward(x, quality)`` with ``quality`` an int / float or a
    per-image tensor, which -- as in the reference (:1476-1480) -- is converted to the compression factor IN PLACE.
    The whole round trip (colour transform, 4:2:0, DCT, quantise, round, inverse) is one HIP kernel, one wavefront per
    16x16 MCU.  Forward only: the reference trains with ``DiffJPEG()`` inside ``torch.no_grad`` data preparation."""

    def __init__(self, differentiable: bool = False) -> None:
        super().__init__()
        self.differentiable = differentiable
        L = A.lib()
        t = np.zeros(L.srganfd_diff_jpeg_table_floats(), dtype=np.float32)
        A.check(L.srganfd_diff_jpeg_tables(t.ctypes.data), "diff_jpeg_tables")
        self.register_buffer("tables", torch.from_numpy(t), persistent=False)

    def forward(self, x: Tensor, quality: Union[int, float, Tensor]) -> Tensor:
        _need_gpu(x, "DiffJPEG")
        xx = x.detach().contiguous().float()
        b, c, h, w = xx.shape
        if isinstance(quality, (int, float)):
            q = 5000. / quality if quality < 50 else 200. - quality * 2
            fac, is_factor = torch.full((b,), q / 100., dtype=torch.float32, device=xx.device), 1
        else:
            if quality.dtype != torch.float32 or not quality.is_cuda or not quality.is_contiguous() or quality.numel() != b:
                raise A.SrganfdError("DiffJPEG: quality tensor must be a contiguous float32 GPU tensor with one entry per image")
            fac, is_factor = quality, 0
        tables = self.tables if self.tables.device == xx.device else self.tables.to(xx.device)
        out = torch.empty_like(xx)
        A.check(A.lib().srganfd_diff_jpeg(xx.data_ptr(), b, c, h, w, fac.data_ptr(), is_factor, 1 if self.differentiable else 0,
                                          tables.data_ptr(), out.data_ptr(), A.stream_ptr()), "diff_jpeg")
        return out
