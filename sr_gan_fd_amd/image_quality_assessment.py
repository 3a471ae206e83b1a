"""PSNR and SSIM of the validation loop (reference: BSRGAN/image_quality_assessment.py:361-418 and :420-532, used at
train_bsrgan.py:545-590)."""
from __future__ import annotations

import numpy as np
import torch
from torch import Tensor, nn

from . import _abi as A


class PSNR(nn.Module):
    """Same constructor and call as the reference's ``PSNR(crop_border, only_test_y_channel)``: inputs (N,C,H,W) RGB in
    [0,1]; returns the per-image PSNR in dB as a float64 tensor of shape (N,).  One fused HIP pass (border crop, BT.601
    luma, fp64 squared-error reduction) instead of the reference's slice / matmul / cast / mean chain."""

    def __init__(self, crop_border: int, only_test_y_channel: bool) -> None:
        super().__init__()
        self.crop_border = crop_border
        self.only_test_y_channel = only_test_y_channel

    def forward(self, raw_tensor: Tensor, dst_tensor: Tensor) -> Tensor:
        assert raw_tensor.shape == dst_tensor.shape, \
            f"Supplied images have different sizes {str(raw_tensor.shape)} and {str(dst_tensor.shape)}"
        if not raw_tensor.is_cuda:
            raise A.SrganfdError("PSNR: tensors must be on the GPU (the HIP library is the product; no CPU fallback)")
        a, b = raw_tensor.detach().contiguous().float(), dst_tensor.detach().contiguous().float()
        n, c, h, w = a.shape
        out = torch.empty(n, dtype=torch.float64, device=a.device)
        ws = torch.empty(n * 64, dtype=torch.float64, device=a.device)
        A.check(A.lib().srganfd_psnr(a.data_ptr(), b.data_ptr(), n, c, h, w, self.crop_border, 1 if self.only_test_y_channel else 0,
                                     out.data_ptr(), ws.data_ptr(), A.stream_ptr()), "psnr")
        return out


def gaussian_kernel_1d(window_size: int, sigma: float) -> np.ndarray:
    """The filter the reference obtains from ``cv2.getGaussianKernel(window_size, sigma)`` (OpenCV is not a dependency
    here): OpenCV's documented formula G_i = alpha * exp(-(i - (ksize-1)/2)^2 / (2 sigma^2)) with sum(G) = 1, and
    sigma = 0.3*((ksize-1)*0.5 - 1) + 0.8 when a non-positive sigma is passed.  fp64 column vector (ksize, 1)."""
    if sigma <= 0:
        sigma = 0.3 * ((window_size - 1) * 0.5 - 1) + 0.8
    x = np.arange(window_size, dtype=np.float64) - (window_size - 1) * 0.5
    g = np.exp(-(x * x) / (2.0 * sigma * sigma))
    return (g / g.sum()).reshape(window_size, 1)


class SSIM(nn.Module):
    """Same constructor (including the reference's ``only_only_test_y_channel`` spelling) and call as the reference's
    ``SSIM`` (image_quality_assessment.py:497-532): inputs (N,C,H,W) RGB in [0,1]; returns the per-image SSIM as a float32
    tensor of shape (N,).  One fused HIP pass per image tile (crop, BT.601 luma, the five fp64 window moments, the SSIM
    map and its mean) instead of five grouped conv2d calls over fp64 copies."""

    def __init__(self, crop_border: int, only_only_test_y_channel: bool, window_size: int = 11, gaussian_sigma: float = 1.5) -> None:
        super().__init__()
        self.crop_border = crop_border
        self.only_test_y_channel = only_only_test_y_channel
        self.window_size = window_size
        g = gaussian_kernel_1d(window_size, gaussian_sigma)
        self.gaussian_kernel_window = np.outer(g, g.transpose())
        self._window_dev = None

    def forward(self, raw_tensor: Tensor, dst_tensor: Tensor) -> Tensor:
        assert raw_tensor.shape == dst_tensor.shape, \
            f"Supplied images have different sizes {str(raw_tensor.shape)} and {str(dst_tensor.shape)}"
        if not raw_tensor.is_cuda:
            raise A.SrganfdError("SSIM: tensors must be on the GPU (the HIP library is the product; no CPU fallback)")
        a, b = raw_tensor.detach().contiguous().float(), dst_tensor.detach().contiguous().float()
        n, c, h, w = a.shape
        if self._window_dev is None or self._window_dev.device != a.device:
            self._window_dev = torch.from_numpy(np.ascontiguousarray(self.gaussian_kernel_window, dtype=np.float64)).to(a.device)
        y = 1 if self.only_test_y_channel else 0
        L = A.lib()
        nws = L.srganfd_ssim_workspace_doubles(n, c, h, w, self.crop_border, y, self.window_size)
        out = torch.empty(n, dtype=torch.float32, device=a.device)
        ws = torch.empty(max(int(nws), 1), dtype=torch.float64, device=a.device)
        A.check(L.srganfd_ssim(a.data_ptr(), b.data_ptr(), n, c, h, w, self.crop_border, y, self._window_dev.data_ptr(), self.window_size,
                               out.data_ptr(), ws.data_ptr(), A.stream_ptr()), "ssim")
        return out
