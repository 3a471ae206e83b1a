"""Device-side mirror of the data-side helpers the train loops call on GPU tensors (reference: BSRGAN/imgproc.py for
``random_crop``; Real_ESRGAN/imgproc.py for the on-device degradation stages -- SURVEY 8f N4: ``filter2d_torch``,
``USMSharp``, ``DiffJPEG``, the noise stages, ``degradation_process``).  The CPU-side pieces of that file (kernel synthesis with numpy / scipy, cv2 image I/O)
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


def image_to_tensor_u8(images_u8: Tensor, top: int = 0, left: int = 0, size=None, bgr: bool = True, range_norm: bool = False) -> Tensor:
    """The reference's per-image host ingest for a whole batch on the device: ``cv2.imread(...).astype(np.float32) / 255.``
    (dataset.py:66), the crop window, ``cv2.cvtColor(..., COLOR_BGR2RGB)`` (:81) and ``image_to_tensor(image, range_norm, False)``
    (imgproc.py:331-358: HWC -> CHW, optional [0, 1] -> [-1, 1]) in one kernel.  ``images_u8``: (N, H, W, 3) uint8 on the GPU, as decoded
    (BGR when ``bgr``); returns (N, 3, h, w) fp32.  The batch crosses PCIe as bytes -- a quarter of the fp32 tensors the reference copies."""
    if images_u8.dtype != torch.uint8 or images_u8.dim() != 4 or images_u8.shape[-1] != 3:
        raise A.SrganfdError("image_to_tensor_u8 takes (N, H, W, 3) uint8 images")
    _need_gpu(images_u8, "image_to_tensor_u8")
    n, h, w, _ = images_u8.shape
    ph, pw = (h - top, w - left) if size is None else ((size, size) if isinstance(size, int) else tuple(size))
    src = images_u8.contiguous()
    out = torch.empty(n, 3, ph, pw, dtype=torch.float32, device=src.device)
    A.check(A.lib().srganfd_u8hwc_to_nchw(src.data_ptr(), out.data_ptr(), n, h, w, top, left, ph, pw, 1 if bgr else 0, 255.0, A.stream_ptr()), "u8hwc_to_nchw")
    return out.mul_(2.0).sub_(1.0) if range_norm else out


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

    def _taps(self, device) -> Tuple[Tensor, int]:
        """(filter operand, separable flag).  The constructor's kernel is an outer product, so the two passes run as a
        horizontal + a vertical 1-D filter (2k instead of k*k multiply-adds per pixel); a ``kernel`` buffer that was replaced
        by something of higher rank is detected (checked once per buffer version) and run as the full 2-D filter."""
        key = (self.kernel.data_ptr(), self.kernel._version, str(device))
        if getattr(self, "_taps_key", None) != key:
            k2 = self.kernel.detach().double().cpu().reshape(self.radius, self.radius)
            mid = self.radius // 2
            sep = None
            if float(k2[mid, mid]) != 0.0:
                col, row = k2[:, mid] / k2[mid, mid], k2[mid, :]
                if float((torch.outer(col, row) - k2).abs().max()) <= 1e-6 * float(k2.abs().max()):    # fp32 rounding of the stored outer product is ~2e-7
                    sep = torch.cat([col, row]).float()
            self._taps_cache = ((sep, 1) if sep is not None else (self.kernel.detach().float().contiguous(), 0))
            self._taps_cache = (self._taps_cache[0].to(device), self._taps_cache[1])
            self._taps_key = key
        return self._taps_cache

    def forward(self, x: Tensor, weight: float, threshold: int) -> Tensor:
        _need_gpu(x, "USMSharp")
        xx = x.detach().contiguous().float()
        b, c, h, w = xx.shape
        kk, separable = self._taps(xx.device)
        out = torch.empty_like(xx)
        ws = torch.empty(2 * xx.numel(), dtype=torch.float32, device=xx.device)
        A.check(A.lib().srganfd_usm_sharp(xx.data_ptr(), kk.data_ptr(), separable, b, c, h, w, self.radius, float(weight), float(threshold),
                                          out.data_ptr(), ws.data_ptr(), A.stream_ptr()), "usm_sharp")
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


_RESIZE_MODES = {"area": 0, "bilinear": 1, "bicubic": 2}

# Where the random draws of the noise / JPEG-quality stages are made.  None: on the image's device (the reference's
# behaviour on a GPU).  "cpu": drawn from torch's CPU generator and copied over -- a run seeded with torch.manual_seed then
# consumes exactly the stream the reference consumes when it runs on the CPU (parity tests, device-independent replays).
DRAW_DEVICE = None


def _draw(fn, *shape, device):
    return fn(*shape, dtype=torch.float32, device=DRAW_DEVICE or device).to(device)


def _poisson(rate: Tensor) -> Tensor:
    return torch.poisson(rate.to(DRAW_DEVICE)).to(rate.device) if DRAW_DEVICE else torch.poisson(rate)


def interpolate(x: Tensor, size=None, scale_factor=None, mode: str = "bilinear") -> Tensor:
    """``torch.nn.functional.interpolate`` as degradation_process calls it (Real_ESRGAN/imgproc.py:2374, :2415-2418,
    :2440-2442, :2454-2456): modes "area" / "bilinear" / "bicubic", align_corners unset; ``scale_factor=`` sizes the output
    as floor(in * scale) and maps coordinates with 1 / scale_factor, ``size=`` with in / out -- torch's rules."""
    _need_gpu(x, "interpolate")
    if mode not in _RESIZE_MODES:
        raise ValueError(f"interpolate: mode {mode!r} is not one of {sorted(_RESIZE_MODES)}")
    if (size is None) == (scale_factor is None):
        raise ValueError("only one of size or scale_factor should be defined")
    xx = x.detach().contiguous().float()
    b, c, h, w = xx.shape
    if size is not None:
        oh, ow = (size, size) if isinstance(size, int) else size
        rs_h = rs_w = 0.0
    else:
        sf = (scale_factor, scale_factor) if isinstance(scale_factor, (int, float)) else tuple(scale_factor)
        oh, ow = int(np.floor(float(h * sf[0]))), int(np.floor(float(w * sf[1])))
        rs_h, rs_w = float(np.float32(1.0 / sf[0])), float(np.float32(1.0 / sf[1]))
    out = torch.empty(b, c, int(oh), int(ow), dtype=torch.float32, device=xx.device)
    A.check(A.lib().srganfd_resize(xx.data_ptr(), b * c, h, w, int(oh), int(ow), _RESIZE_MODES[mode], rs_h, rs_w, out.data_ptr(), A.stream_ptr()), "resize")
    return out


def _per_image(v, b: int, device) -> Tensor:
    if isinstance(v, (float, int)):
        return torch.full((b,), float(v), dtype=torch.float32, device=device)
    return v.detach().to(device=device, dtype=torch.float32).reshape(b).contiguous()


def _add_gaussian_noise_torch(image: Tensor, sigma=10.0, clip: bool = True, rounds: bool = False, gray_noise=0) -> Tensor:
    """imgproc._add_gaussian_noise_torch (Real_ESRGAN/imgproc.py:970-998 over :832-866): the draws come from torch's
    generator on the image's device in the reference's order (the shared (h, w) grey field first, if any image asks for
    grey noise, then the colour field); scaling, grey / colour mixing, the add and the clip are one HIP pass."""
    _need_gpu(image, "_add_gaussian_noise_torch")
    x = image.detach().contiguous().float()
    b, c, h, w = x.shape
    sg = _per_image(sigma, b, x.device)
    if isinstance(gray_noise, (float, int)):
        cal_gray, gray = gray_noise > 0, _per_image(gray_noise, b, x.device)
    else:
        gray = _per_image(gray_noise, b, x.device)
        cal_gray = bool(torch.sum(gray) > 0)
    n_gray = _draw(torch.randn, h, w, device=x.device) if cal_gray else None
    n_color = _draw(torch.randn, b, c, h, w, device=x.device)
    return gaussian_noise_apply(x, n_color, n_gray, sg, gray, clip, rounds)


def gaussian_noise_apply(image: Tensor, randn_color: Tensor, randn_gray_hw, sigma: Tensor, gray_flag: Tensor, clip: bool, rounds: bool) -> Tensor:
    """the deterministic part of the Gaussian-noise stage on given draws (srganfd_gaussian_noise)"""
    b, c, h, w = image.shape
    out = torch.empty_like(image)
    A.check(A.lib().srganfd_gaussian_noise(image.data_ptr(), randn_color.data_ptr(), randn_gray_hw.data_ptr() if randn_gray_hw is not None else None,
                                           sigma.data_ptr(), gray_flag.data_ptr(), b, c, h, w, int(clip), int(rounds), out.data_ptr(), A.stream_ptr()),
            "gaussian_noise")
    return out


def random_add_gaussian_noise_torch(image: Tensor, sigma_range: tuple = (0, 1.0), gray_prob: int = 0, clip: bool = True, rounds: bool = False) -> Tensor:
    """imgproc.random_add_gaussian_noise_torch (Real_ESRGAN/imgproc.py:1032-1060 over :922-943): per-image sigma and grey
    flag drawn with torch.rand in the reference's order, then ``_add_gaussian_noise_torch``."""
    b = image.size(0)
    sigma = _draw(torch.rand, b, device=image.device) * (sigma_range[1] - sigma_range[0]) + sigma_range[0]
    gray_noise = (_draw(torch.rand, b, device=image.device) < gray_prob).float()
    return _add_gaussian_noise_torch(image, sigma, clip, rounds, gray_noise)


def poisson_noise_prepare(image: Tensor, want_gray: bool):
    """8-bit rounded image (and grey image), and vals = 2^ceil(log2(#distinct levels)) per image (imgproc.py:892-910)"""
    b, c, h, w = image.shape
    img_q = torch.empty_like(image)
    gray_q = torch.empty(b, 1, h, w, dtype=torch.float32, device=image.device) if want_gray else None
    vals = torch.empty(b, dtype=torch.float32, device=image.device)
    vals_gray = torch.empty(b, dtype=torch.float32, device=image.device) if want_gray else None
    ws = torch.empty(b * 512, dtype=torch.int32, device=image.device)
    A.check(A.lib().srganfd_poisson_prepare(image.data_ptr(), b, c, h, w, int(want_gray), img_q.data_ptr(), gray_q.data_ptr() if want_gray else None,
                                            vals.data_ptr(), vals_gray.data_ptr() if want_gray else None, ws.data_ptr(), A.stream_ptr()), "poisson_prepare")
    return img_q, gray_q, vals, vals_gray


def poisson_noise_apply(image, img_q, gray_q, pois, pois_gray, vals, vals_gray, scale, gray_flag, clip: bool, rounds: bool) -> Tensor:
    """the deterministic part of the Poisson-noise stage on given draws (srganfd_poisson_apply)"""
    b, c, h, w = image.shape
    out = torch.empty_like(image)
    P = lambda t: t.data_ptr() if t is not None else None
    A.check(A.lib().srganfd_poisson_apply(image.data_ptr(), img_q.data_ptr(), P(gray_q), pois.data_ptr(), P(pois_gray), vals.data_ptr(), P(vals_gray),
                                          scale.data_ptr(), P(gray_flag), b, c, h, w, int(clip), int(rounds), out.data_ptr(), A.stream_ptr()), "poisson_apply")
    return out


def _add_poisson_noise_torch(image: Tensor, scale=1.0, clip: bool = True, rounds: bool = False, gray_noise=0) -> Tensor:
    """imgproc._add_poisson_noise_torch (Real_ESRGAN/imgproc.py:1001-1029 over :869-919): torch.poisson draws (grey first, as
    in the reference) on rates prepared by one HIP pass, everything after the draws in another."""
    _need_gpu(image, "_add_poisson_noise_torch")
    x = image.detach().contiguous().float()
    b, c, h, w = x.shape
    if isinstance(gray_noise, (float, int)):
        cal_gray, gray = gray_noise > 0, _per_image(gray_noise, b, x.device)
    else:
        gray = _per_image(gray_noise, b, x.device)
        cal_gray = bool(torch.sum(gray) > 0)
    img_q, gray_q, vals, vals_gray = poisson_noise_prepare(x, cal_gray)
    pois_gray = _poisson(gray_q * vals_gray.view(b, 1, 1, 1)) if cal_gray else None
    pois = _poisson(img_q * vals.view(b, 1, 1, 1))
    return poisson_noise_apply(x, img_q, gray_q, pois, pois_gray, vals, vals_gray, _per_image(scale, b, x.device), gray if cal_gray else None, clip, rounds)


def random_add_poisson_noise_torch(image: Tensor, scale_range: tuple = (0, 1.0), gray_prob: int = 0, clip: bool = True, rounds: bool = False) -> Tensor:
    """imgproc.random_add_poisson_noise_torch (Real_ESRGAN/imgproc.py:1063-1089 over :946-967)"""
    b = image.size(0)
    scale = _draw(torch.rand, b, device=image.device) * (scale_range[1] - scale_range[0]) + scale_range[0]
    gray_noise = (_draw(torch.rand, b, device=image.device) < gray_prob).float()
    return _add_poisson_noise_torch(image, scale, clip, rounds, gray_noise)


def quantize_u8(x: Tensor) -> Tensor:
    """clamp(round(x * 255), 0, 255) / 255 -- the last line of degradation_process (Real_ESRGAN/imgproc.py:2460)"""
    _need_gpu(x, "quantize_u8")
    xx = x.detach().contiguous().float()
    out = torch.empty_like(xx)
    A.check(A.lib().srganfd_quantize_u8(xx.data_ptr(), out.data_ptr(), xx.numel(), A.stream_ptr()), "quantize_u8")
    return out


def _jpeg_quality(out: Tensor, jpeg_range) -> Tensor:
    """quality = out.new_zeros(b).uniform_(*range) (imgproc.py:2393-2394)"""
    return torch.zeros(out.size(0), dtype=torch.float32, device=DRAW_DEVICE or out.device).uniform_(*jpeg_range).to(out.device)


def degradation_process(gt: Tensor, gaussian_kernel1: Tensor, gaussian_kernel2: Tensor, sinc_kernel: Tensor, upscale_factor: int,
                        degradation_process_parameters_dict: dict, jpeg_operation: nn.Module = None, usm_sharpener: nn.Module = None):
    """imgproc.degradation_process (Real_ESRGAN/imgproc.py:2323-2462): the second-order degradation of a GT batch on the
    GPU -- [sharpen] blur, random resize, Gaussian-or-Poisson noise, JPEG; blur, resize, noise; then resize + sinc filter
    and JPEG in a random order; 8-bit quantisation.  Host-side draws (numpy / ``random``) are made in the reference's
    order, so a seeded run takes the same branches; every stage is a HIP kernel of this library.  Returns
    ``(gt_usm, gt, lr)``.  Differences from the reference, both where it cannot run as written: the sharpener is called
    as ``usm_sharpener(gt, 0.5, 10)`` (the reference passes no weight / threshold to a forward that requires them;
    0.5 / 10 are its numpy twin's defaults, :1500), and a skipped first blur passes the image on (the reference would hit
    an unbound ``out``; its configs use probability 1.0)."""
    P = degradation_process_parameters_dict
    image_height, image_width = gt.size()[2:4]
    gt_usm = gt
    if usm_sharpener is not None:
        gt_usm = usm_sharpener(gt, 0.5, 10)
    out = gt_usm
    # first degradation: blur, resize, noise, JPEG
    if np.random.uniform() <= P["first_blur_probability"]:
        out = filter2d_torch(gt_usm, gaussian_kernel1)
    updown_type = random.choices(["up", "down", "keep"], P["resize_probability1"])[0]
    if updown_type == "up":
        scale = np.random.uniform(1, P["resize_range1"][1])
    elif updown_type == "down":
        scale = np.random.uniform(P["resize_range1"][0], 1)
    else:
        scale = 1
    mode = random.choice(["area", "bilinear", "bicubic"])
    out = interpolate(out, scale_factor=scale, mode=mode)
    if np.random.uniform() < P["gaussian_noise_probability1"]:
        out = random_add_gaussian_noise_torch(image=out, sigma_range=P["noise_range1"], clip=True, rounds=False, gray_prob=P["gray_noise_probability1"])
    else:
        out = random_add_poisson_noise_torch(image=out, scale_range=P["poisson_scale_range1"], gray_prob=P["gray_noise_probability1"], clip=True,
                                             rounds=False)
    quality = _jpeg_quality(out, P["jpeg_range1"])
    out = jpeg_operation(torch.clamp(out, 0, 1), quality)
    # second degradation: blur, resize, noise
    if np.random.uniform() < P["second_blur_probability"]:
        out = filter2d_torch(out, gaussian_kernel2)
    updown_type = random.choices(["up", "down", "keep"], P["resize_probability2"])[0]
    if updown_type == "up":
        scale = np.random.uniform(1, P["resize_range2"][1])
    elif updown_type == "down":
        scale = np.random.uniform(P["resize_range2"][0], 1)
    else:
        scale = 1
    mode = random.choice(["area", "bilinear", "bicubic"])
    out = interpolate(out, size=(int(image_height / upscale_factor * scale), int(image_width / upscale_factor * scale)), mode=mode)
    if np.random.uniform() < P["gaussian_noise_probability2"]:
        out = random_add_gaussian_noise_torch(image=out, sigma_range=P["noise_range2"], clip=True, rounds=False, gray_prob=P["gray_noise_probability2"])
    else:
        out = random_add_poisson_noise_torch(image=out, scale_range=P["poisson_scale_range2"], gray_prob=P["gray_noise_probability2"], clip=True,
                                             rounds=False)
    final_size = (image_height // upscale_factor, image_width // upscale_factor)
    if np.random.uniform() < 0.5:
        # resize back -> sinc filter -> JPEG
        out = interpolate(out, size=final_size, mode=random.choice(["area", "bilinear", "bicubic"]))
        out = filter2d_torch(out, sinc_kernel)
        quality = _jpeg_quality(out, P["jpeg_range2"])
        out = jpeg_operation(torch.clamp(out, 0, 1), quality)
    else:
        # JPEG -> resize back -> sinc filter
        quality = _jpeg_quality(out, P["jpeg_range2"])
        out = jpeg_operation(torch.clamp(out, 0, 1), quality)
        out = interpolate(out, size=final_size, mode=random.choice(["area", "bilinear", "bicubic"]))
        out = filter2d_torch(out, sinc_kernel)
    lr = quantize_u8(out)
    return gt_usm, gt, lr


# ---- batch augmentation of the Real-ESRGAN loop (train_realesrgan.py:400-404) ------------------------------------------------
def _as_list(v):
    return (v, True) if isinstance(v, list) else ([v], False)


def _crop_rot_flip(t: Tensor, top: int, left: int, ph: int, pw: int, op: int) -> Tensor:
    _need_gpu(t, "augmentation")
    x = t.detach().contiguous().float()
    b, c, h, w = x.shape
    out = torch.empty(b, c, ph, pw, dtype=torch.float32, device=x.device)
    A.check(A.lib().srganfd_crop_rot_flip(x.data_ptr(), out.data_ptr(), b * c, h, w, top, left, ph, pw, op, A.stream_ptr()), "crop_rot_flip")
    return out.to(t.dtype)


def random_crop_torch(gt_images, lr_images, gt_patch_size: int, upscale_factor: int):
    """imgproc.random_crop_torch (Real_ESRGAN/imgproc.py:2081-2155), tensor inputs: one LR window drawn with ``random.randint``
    (row, then column) for every tensor of both lists, the GT window at upscale_factor times its origin."""
    gts, _ = _as_list(gt_images)
    lrs, _ = _as_list(lr_images)
    lh, lw = lrs[0].size()[-2:]
    lps = gt_patch_size // upscale_factor
    lr_top = random.randint(0, lh - lps)
    lr_left = random.randint(0, lw - lps)
    lrs = [_crop_rot_flip(t, lr_top, lr_left, lps, lps, 0) for t in lrs]
    gts = [_crop_rot_flip(t, int(lr_top * upscale_factor), int(lr_left * upscale_factor), gt_patch_size, gt_patch_size, 0) for t in gts]
    return (gts[0] if len(gts) == 1 else gts), (lrs[0] if len(lrs) == 1 else lrs)


def random_rotate_torch(gt_images, lr_images, upscale_factor: int, angles: list, gt_center=None, lr_center=None, rotate_scale_factor: float = 1.0):
    """imgproc.random_rotate_torch (Real_ESRGAN/imgproc.py:2158-2230), tensor inputs: one angle drawn with ``random.choice``;
    the reference rotates with torchvision about the default centre [w // 2, h // 2], which for the multiples of 90 degrees the
    train loops pass (train_realesrgan.py:401) and square even-sized batches is an exact counter-clockwise quarter-turn
    permutation -- the only case with a HIP path."""
    angle = random.choice(angles)
    gts, _ = _as_list(gt_images)
    lrs, _ = _as_list(lr_images)
    if gt_center is not None or lr_center is not None or rotate_scale_factor != 1.0 or angle % 90 != 0:
        raise A.SrganfdError("random_rotate_torch: only default-centre rotations by multiples of 90 degrees have a HIP path")
    op = (angle // 90) % 4
    out = []
    for group in (gts, lrs):
        res = []
        for t in group:
            h, w = t.shape[-2:]
            if op in (1, 3) and (h != w or h % 2):
                raise A.SrganfdError("random_rotate_torch: quarter turns need square, even-sized images")
            res.append(_crop_rot_flip(t, 0, 0, h, w, op))
        out.append(res)
    return (out[0][0] if len(out[0]) == 1 else out[0]), (out[1][0] if len(out[1]) == 1 else out[1])


def _random_flip(gt_images, lr_images, p: float, op: int):
    flip_prob = random.random()
    gts, _ = _as_list(gt_images)
    lrs, _ = _as_list(lr_images)
    if flip_prob > p:
        lrs = [_crop_rot_flip(t, 0, 0, t.shape[-2], t.shape[-1], op) for t in lrs]
        gts = [_crop_rot_flip(t, 0, 0, t.shape[-2], t.shape[-1], op) for t in gts]
    return (gts[0] if len(gts) == 1 else gts), (lrs[0] if len(lrs) == 1 else lrs)


def random_horizontally_flip_torch(gt_images, lr_images, p: float = 0.5):
    """imgproc.random_horizontally_flip_torch (Real_ESRGAN/imgproc.py:2233-2275): flips when ``random.random() > p``"""
    return _random_flip(gt_images, lr_images, p, 4)


def random_vertically_flip_torch(gt_images, lr_images, p: float = 0.5):
    """imgproc.random_vertically_flip_torch (Real_ESRGAN/imgproc.py:2278-2320)"""
    return _random_flip(gt_images, lr_images, p, 5)
