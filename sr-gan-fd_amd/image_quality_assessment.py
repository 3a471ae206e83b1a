"""PSNR of the validation loop (reference: BSRGAN/image_quality_assessment.py:361-418, used at train_bsrgan.py:566)."""
from __future__ import annotations

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
