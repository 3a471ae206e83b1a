"""Device-side mirror of the data-side helpers the train loop calls on GPU tensors (reference: BSRGAN/imgproc.py).

Only what sits on the training / validation path is here; the degradation pipeline is out of scope."""
from __future__ import annotations

import random
from typing import Tuple

import torch
from torch import Tensor

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
