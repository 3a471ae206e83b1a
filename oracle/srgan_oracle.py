"""CPU oracle for the SR-GAN-FD hot path.  TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (torch-CPU tensor ops, fp32 or fp64) of the reference's
generator / discriminator / loss / optimizer arithmetic.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it; the product
path (``sr_gan_fd_amd``) never does and fails loudly when its HIP library is missing.

Pinned by: ``tests/golden/*.npz`` -- vectors captured by importing the reference's own
``BSRGAN/model.py`` / ``ESRGAN/model.py`` in the build container (``tests/golden/make_golden.py``,
which is the only file that touches /root/reference) and checked in ``tests/test_oracle_golden.py``.
Not pinned: VGG-19 content-loss values (torchvision + ImageNet weights are absent from the
reference tree and from this image) -- "parity unpinned" for that one function, see DESIGN.md.

Every function cites the reference file:line (relative to /root/reference) it follows.
Parameters are passed as ``dict[str, Tensor]`` using the reference's ``state_dict`` key names so
the same oracle call works for a reference checkpoint and for the MI355X module.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]

LRELU_SLOPE = 0.2


# ----------------------------------------------------------------------------------------------
# Generator (RRDBNet / BSRGAN)
# ----------------------------------------------------------------------------------------------
def _conv(x: Tensor, P: Params, name: str, stride: int = 1, pad: int = 1) -> Tensor:
    return F.conv2d(x, P[name + ".weight"], P.get(name + ".bias"), stride=stride, padding=pad)


def rdb_forward(x: Tensor, P: Params, prefix: str) -> Tensor:
    """_ResidualDenseBlock.forward -- BSRGAN/model.py:51-62 (same code ESRGAN/model.py:49-60)."""
    feats = [x]
    for k in range(1, 5):
        y = _conv(torch.cat(feats, 1), P, f"{prefix}conv{k}")
        feats.append(F.leaky_relu(y, LRELU_SLOPE))
    out5 = _conv(torch.cat(feats, 1), P, f"{prefix}conv5")
    return out5 * 0.2 + x


def rrdb_forward(x: Tensor, P: Params, prefix: str) -> Tensor:
    """_ResidualResidualDenseBlock.forward -- BSRGAN/model.py:79-88."""
    out = x
    for r in (1, 2, 3):
        out = rdb_forward(out, P, f"{prefix}rdb{r}.")
    return out * 0.2 + x


def count_rrdb(P: Params) -> int:
    n = 0
    while f"trunk.{n}.rdb1.conv1.weight" in P:
        n += 1
    return n


def rrdbnet_forward(x: Tensor, P: Params, upscale_factor: int = 4, clamp: bool = True, unshuffle: int = 1) -> Tensor:
    """BSRGAN._forward_impl -- BSRGAN/model.py:366-381; RRDBNet._forward_impl -- ESRGAN/model.py:208-229.
    ``unshuffle`` = 2 / 4: Real_ESRGAN/model.py:190-204,248-262 below x4 -- the input is pixel-unshuffled (conv1 reads 12 / 48 channels)
    and both nearest-x2 stages run (pass upscale_factor=4 for the stage count; the net factor is 4 / unshuffle).

    x: (N,3,h,w) in [0,1]  ->  (N,3,s*h,s*w), clamped to [0,1].
    """
    if unshuffle > 1:
        x = F.pixel_unshuffle(x, unshuffle)
    out1 = _conv(x, P, "conv1")
    out = out1
    for i in range(count_rrdb(P)):
        out = rrdb_forward(out, P, f"trunk.{i}.")
    out2 = _conv(out, P, "conv2")
    out = out1 + out2
    n_up = {1: 0, 2: 1, 4: 2, 8: 3}[upscale_factor]
    for u in range(1, n_up + 1):
        out = F.interpolate(out, scale_factor=2, mode="nearest")
        out = F.leaky_relu(_conv(out, P, f"upsampling{u}.0"), LRELU_SLOPE)
    out = F.leaky_relu(_conv(out, P, "conv3.0"), LRELU_SLOPE)
    out = _conv(out, P, "conv4")
    if clamp:
        out = torch.clamp(out, 0.0, 1.0)
    return out


# ----------------------------------------------------------------------------------------------
# Spectral norm + U-Net discriminator
# ----------------------------------------------------------------------------------------------
def _normalize(v: Tensor, eps: float) -> Tensor:
    return v / torch.clamp(v.norm(), min=eps)


def spectral_norm_weight(w_orig: Tensor, u: Tensor, v: Tensor, training: bool,
                         eps: float = 1e-12, n_power_iterations: int = 1
                         ) -> Tuple[Tensor, Tensor, Tensor]:
    """torch/nn/utils/spectral_norm.py:62-114 as applied at BSRGAN/model.py:104-132.

    Returns (weight = w_orig / sigma, u_new, v_new).  u, v are treated as constants for autograd;
    gradient reaches w_orig through both the numerator and sigma = u^T W v.
    """
    w_mat = w_orig.reshape(w_orig.shape[0], -1)
    if training:
        with torch.no_grad():
            for _ in range(n_power_iterations):
                v = _normalize(torch.mv(w_mat.t(), u), eps)
                u = _normalize(torch.mv(w_mat, v), eps)
            u = u.clone()
            v = v.clone()
    sigma = torch.dot(u, torch.mv(w_mat, v))
    return w_orig / sigma, u, v


D_SN_LAYERS = ("down_block1", "down_block2", "down_block3", "up_block1", "up_block2", "up_block3",
               "conv2", "conv3")


def discriminator_unet_forward(x: Tensor, P: Params, training: bool = True,
                               update_state: bool = True) -> Tensor:
    """DiscriminatorUNet._forward_impl -- BSRGAN/model.py:141-167 (identical Real_ESRGAN/model.py:79-105).

    P holds conv1/conv4 weight+bias and, per SN layer L, ``L.0.weight_orig``, ``L.0.weight_u``,
    ``L.0.weight_v``.  In training mode the u/v entries of P are advanced in place (one power
    iteration per forward) exactly like the reference's forward-pre-hook does.
    """
    def sn_conv(inp: Tensor, layer: str, stride: int) -> Tensor:
        w, u, v = spectral_norm_weight(P[f"{layer}.0.weight_orig"], P[f"{layer}.0.weight_u"],
                                       P[f"{layer}.0.weight_v"], training)
        if training and update_state:
            P[f"{layer}.0.weight_u"] = u
            P[f"{layer}.0.weight_v"] = v
        return F.leaky_relu(F.conv2d(inp, w, None, stride=stride, padding=1), LRELU_SLOPE)

    out1 = _conv(x, P, "conv1")
    down1 = sn_conv(out1, "down_block1", 2)
    down2 = sn_conv(down1, "down_block2", 2)
    down3 = sn_conv(down2, "down_block3", 2)
    down3 = F.interpolate(down3, scale_factor=2, mode="bilinear", align_corners=False)
    up1 = sn_conv(down3, "up_block1", 1) + down2
    up1 = F.interpolate(up1, scale_factor=2, mode="bilinear", align_corners=False)
    up2 = sn_conv(up1, "up_block2", 1) + down1
    up2 = F.interpolate(up2, scale_factor=2, mode="bilinear", align_corners=False)
    up3 = sn_conv(up2, "up_block3", 1) + out1
    out = sn_conv(up3, "conv2", 1)
    out = sn_conv(out, "conv3", 1)
    return _conv(out, P, "conv4")


# ----------------------------------------------------------------------------------------------
# VGG-19 feature taps / content loss  (parity unpinned -- third-party torchvision, see header)
# ----------------------------------------------------------------------------------------------
VGG19_CFG: Sequence = (64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M",
                       512, 512, 512, 512, "M", 512, 512, 512, 512)  # features[0:35]


def vgg19_feature_layers() -> List[Tuple[str, int, int]]:
    """(kind, index-in-features, channels) list of torchvision vgg19().features[0:36]."""
    layers, idx, cin = [], 0, 3
    for v in VGG19_CFG:
        if v == "M":
            layers.append(("pool", idx, cin))
            idx += 1
        else:
            layers.append(("conv", idx, v))
            layers.append(("relu", idx + 1, v))
            idx += 2
            cin = v
    return layers


def vgg19_taps(x: Tensor, P: Params, nodes: Sequence[str], taps_post_relu: bool = True
               ) -> Dict[str, Tensor]:
    """torchvision vgg19().features cut at ``features.N`` conv nodes -- BSRGAN/model.py:522-524,545-546.

    With ``taps_post_relu`` the tap of a conv that is followed by ``ReLU(inplace=True)`` is
    observed after the in-place ReLU (the graph keeps executing until the last requested node),
    while the last requested node is observed pre-ReLU (later nodes are pruned) -- SURVEY 8a/A7.
    P keys: ``features.{idx}.weight`` / ``.bias``.
    """
    want = [int(n.split(".")[1]) for n in nodes]
    last = max(want)
    out: Dict[str, Tensor] = {}
    h = x
    for kind, idx, _ in vgg19_feature_layers():
        if idx > last:
            break
        if kind == "conv":
            h = F.conv2d(h, P[f"features.{idx}.weight"], P[f"features.{idx}.bias"], padding=1)
            if idx in want:
                if idx == last or not taps_post_relu:
                    out[f"features.{idx}"] = h
                else:
                    out[f"features.{idx}"] = F.relu(h)
        elif kind == "relu":
            h = F.relu(h)
        else:
            h = F.max_pool2d(h, 2, 2)
    return out


def content_loss(sr: Tensor, gt: Tensor, P: Params, nodes: Sequence[str], mean: Sequence[float],
                 std: Sequence[float], taps_post_relu: bool = True) -> Tensor:
    """ContentLoss.forward -- BSRGAN/model.py:536-554.  Returns a DETACHED (1, len(nodes)) tensor
    (``torch.Tensor([losses])`` at :552 builds a fresh leaf)."""
    m = torch.tensor(mean, dtype=sr.dtype).view(1, -1, 1, 1)
    s = torch.tensor(std, dtype=sr.dtype).view(1, -1, 1, 1)
    with torch.no_grad():
        fs = vgg19_taps((sr - m) / s, P, nodes, taps_post_relu)
        fg = vgg19_taps((gt - m) / s, P, nodes, taps_post_relu)
        losses = [float(F.l1_loss(fs[n], fg[n])) for n in nodes]
    return torch.tensor([losses], dtype=torch.float32)


# ----------------------------------------------------------------------------------------------
# Losses, PSNR
# ----------------------------------------------------------------------------------------------
def l1_mean(a: Tensor, b: Tensor) -> Tensor:
    """nn.L1Loss() mean -- BSRGAN/train_bsrgan.py:297,450."""
    return (a - b).abs().mean()


def bce_with_logits_mean(logits: Tensor, target: float) -> Tensor:
    """nn.BCEWithLogitsLoss() mean against a constant label map -- train_bsrgan.py:301,403-404,417,427,452.
    loss = max(x,0) - x*t + log(1+exp(-|x|))."""
    x = logits
    return (torch.clamp(x, min=0) - x * target + torch.log1p(torch.exp(-x.abs()))).mean()


def psnr_y(raw: Tensor, dst: Tensor, crop_border: int = 0, only_test_y_channel: bool = True) -> Tensor:
    """_psnr_torch -- BSRGAN/image_quality_assessment.py:361-395 with rgb_to_ycbcr_torch
    (BSRGAN/imgproc.py:742-770, only_use_y_channel=True branch).  Inputs (N,3,H,W) in [0,1], RGB."""
    if crop_border > 0:
        raw = raw[:, :, crop_border:-crop_border, crop_border:-crop_border]
        dst = dst[:, :, crop_border:-crop_border, crop_border:-crop_border]
    if only_test_y_channel:
        w = torch.tensor([[65.481], [128.553], [24.966]], dtype=raw.dtype)

        def to_y(t: Tensor) -> Tensor:
            y = torch.matmul(t.permute(0, 2, 3, 1), w).permute(0, 3, 1, 2) + 16.0
            return y / 255.0
        raw, dst = to_y(raw), to_y(dst)
    raw = raw.to(torch.float64)
    dst = dst.to(torch.float64)
    mse = torch.mean((raw * 255.0 - dst * 255.0) ** 2 + 1e-8, dim=[1, 2, 3])
    return 10 * torch.log10(255.0 ** 2 / mse)


def gaussian_window(window_size: int = 11, sigma: float = 1.5):
    """SSIM.__init__ -- BSRGAN/image_quality_assessment.py:520-521: outer product of cv2.getGaussianKernel(ws, sigma) with
    itself.  OpenCV (a third-party dependency absent from this image; any 4.x) documents that kernel as
    G_i = alpha * exp(-(i-(ksize-1)/2)^2 / (2 sigma^2)), sum(G) = 1, computed in fp64 -- restated here."""
    import numpy as np
    x = np.arange(window_size, dtype=np.float64) - (window_size - 1) / 2.0
    g = np.exp(-(x ** 2) / (2.0 * sigma ** 2))
    g = g / g.sum()
    return np.outer(g, g)


def ssim(raw: Tensor, dst: Tensor, crop_border: int = 0, only_test_y_channel: bool = True, window=None) -> Tensor:
    """_ssim_single_torch + _ssim_torch -- BSRGAN/image_quality_assessment.py:420-494: crop, fp32 luma, fp64 x255, five
    valid-padding grouped filterings with the 2-D window, SSIM map, mean over (C,H,W), cast to fp32."""
    import numpy as np
    window = gaussian_window() if window is None else window
    ws = window.shape[0]
    if crop_border > 0:
        raw = raw[:, :, crop_border:-crop_border, crop_border:-crop_border]
        dst = dst[:, :, crop_border:-crop_border, crop_border:-crop_border]
    if only_test_y_channel:
        wy = torch.tensor([[65.481], [128.553], [24.966]], dtype=raw.dtype)

        def to_y(t: Tensor) -> Tensor:
            return (torch.matmul(t.permute(0, 2, 3, 1), wy).permute(0, 3, 1, 2) + 16.0) / 255.0
        raw, dst = to_y(raw), to_y(dst)
    a = raw.to(torch.float64).numpy() * 255.0
    b = dst.to(torch.float64).numpy() * 255.0
    oh, ow = a.shape[2] - ws + 1, a.shape[3] - ws + 1

    def filt(t):
        acc = np.zeros(t.shape[:2] + (oh, ow), dtype=np.float64)
        for ky in range(ws):
            for kx in range(ws):
                acc += window[ky, kx] * t[:, :, ky:ky + oh, kx:kx + ow]
        return acc
    c1, c2 = (0.01 * 255.0) ** 2, (0.03 * 255.0) ** 2
    ma, mb = filt(a), filt(b)
    va, vb, cab = filt(a * a) - ma ** 2, filt(b * b) - mb ** 2, filt(a * b) - ma * mb
    m = ((2 * ma * mb + c1) * (2 * cab + c2)) / ((ma ** 2 + mb ** 2 + c1) * (va + vb + c2))
    return torch.from_numpy(m.mean(axis=(1, 2, 3))).float()


def random_crop(gt: Tensor, lr: Tensor, gt_image_size: int, upscale_factor: int, rng=None) -> Tuple[Tensor, Tensor]:
    """random_crop -- BSRGAN/imgproc.py:846-886: ONE (top, left) for the whole batch from Python's `random` stream
    (randint for the row first, then the column), LR window at the integer-divided position; outputs take lr's dtype."""
    import random as _random
    rng = rng or _random
    h, w = gt.shape[2], gt.shape[3]
    top = rng.randint(0, h - gt_image_size)
    left = rng.randint(0, w - gt_image_size)
    lt, ll, ls = top // upscale_factor, left // upscale_factor, gt_image_size // upscale_factor
    return (gt[:, :, top:top + gt_image_size, left:left + gt_image_size].to(lr.dtype).clone(),
            lr[:, :, lt:lt + ls, ll:ll + ls].clone())


# ----------------------------------------------------------------------------------------------
# Optimizer / EMA  (torch.optim.Adam single-tensor maths; swa_utils.AveragedModel)
# ----------------------------------------------------------------------------------------------
class AdamState:
    def __init__(self, params: Params, names: Sequence[str]):
        self.step = 0
        self.m = {k: torch.zeros_like(params[k]) for k in names}
        self.v = {k: torch.zeros_like(params[k]) for k in names}


def adam_step(params: Params, grads: Params, st: AdamState, lr: float,
              betas: Tuple[float, float], eps: float, weight_decay: float = 0.0) -> None:
    """torch.optim.Adam (amsgrad=False, maximize=False) as configured at train_bsrgan.py:311-323
    (eps=1e-4, betas=(0.9,0.999)) / train_rrdbnet.py:196-203 (eps=1e-8, betas=(0.9,0.99))."""
    st.step += 1
    b1, b2 = betas
    bc1 = 1 - b1 ** st.step
    bc2 = 1 - b2 ** st.step
    for k, g in grads.items():
        if g is None:
            continue
        p = params[k]
        if weight_decay != 0:
            g = g + weight_decay * p
        st.m[k].mul_(b1).add_(g, alpha=1 - b1)
        st.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (st.v[k].sqrt() / math.sqrt(bc2)).add_(eps)
        p.data.addcdiv_(st.m[k], denom, value=-(lr / bc1))


def ema_update(ema: Params, params: Params, n_averaged: int, decay: float) -> int:
    """AveragedModel.update_parameters with the reference's avg_fn
    ``(1-decay)*ema + decay*param`` -- train_bsrgan.py:290-291 (note: weight ``decay`` on the NEW
    parameters; kept as is).  First call copies (torch/optim/swa_utils.py)."""
    for k in params:
        if n_averaged == 0:
            ema[k] = params[k].detach().clone()
        else:
            ema[k] = (1 - decay) * ema[k] + decay * params[k].detach()
    return n_averaged + 1


# ----------------------------------------------------------------------------------------------
# Training iterations
# ----------------------------------------------------------------------------------------------
def _leafify(P: Params, names: Sequence[str]) -> None:
    for k in names:
        P[k] = P[k].detach().requires_grad_(True)


def g_param_names(P: Params) -> List[str]:
    return [k for k in P if k.endswith(".weight") or k.endswith(".bias")]


def d_param_names(P: Params) -> List[str]:
    return [k for k in P if k.endswith(".weight") or k.endswith(".bias") or k.endswith("weight_orig")]


def g_only_step(G: Params, opt: AdamState, lr_img: Tensor, gt: Tensor, *, upscale: int, lr: float,
                betas: Tuple[float, float], eps: float, loss_weight: float = 1.0) -> Tuple[float, Tensor]:
    """Generator-only iteration -- ESRGAN/train_rrdbnet.py:244-267, BSRGAN/train_bsrnet.py:244-272.
    (autocast / GradScaler are inert on CPU: SURVEY 8 preamble.)  Returns (loss, sr)."""
    names = g_param_names(G)
    _leafify(G, names)
    sr = rrdbnet_forward(lr_img, G, upscale)
    loss = loss_weight * l1_mean(sr, gt)
    grads = torch.autograd.grad(loss, [G[k] for k in names])
    with torch.no_grad():
        adam_step(G, dict(zip(names, grads)), opt, lr, betas, eps)
    for k in names:
        G[k] = G[k].detach()
    return float(loss), sr.detach()


def gan_step(G: Params, D: Params, g_opt: AdamState, d_opt: AdamState, lr_img: Tensor, gt: Tensor, *,
             upscale: int, g_lr: float, d_lr: float, betas: Tuple[float, float], eps: float,
             pixel_weight: float, content_weight: float, adversarial_weight: float,
             content_fn=None, train_generator: bool = True, d_forward=None) -> Dict[str, float]:
    """One GAN iteration -- BSRGAN/train_bsrgan.py:387-483 (same statements in A-ESRGAN/train_aesrgan.py:396-483;
    ``d_forward(x, D, training=True)`` selects the discriminator, default DiscriminatorUNet), exact order:
    D(gt) fwd+bwd, G fwd, D(sr.detach()) fwd+bwd (accumulate), D step, freeze D, pixel/content/adv
    with the UPDATED D (SN u/v advance a third time), G bwd + step.  content_fn(sr, gt) returns the
    detached (1,5) tensor or None (-> 0)."""
    gn, dn = g_param_names(G), d_param_names(D)
    discriminator_unet_forward = d_forward or globals()["discriminator_unet_forward"]
    _leafify(D, dn)
    _leafify(G, gn)
    # ---- D step ----
    gt_out = discriminator_unet_forward(gt, D, training=True)
    d_loss_hr = bce_with_logits_mean(gt_out, 1.0)
    g_hr = torch.autograd.grad(d_loss_hr, [D[k] for k in dn])
    sr = rrdbnet_forward(lr_img, G, upscale)
    sr_out = discriminator_unet_forward(sr.detach().clone(), D, training=True)
    d_loss_sr = bce_with_logits_mean(sr_out, 0.0)
    g_sr = torch.autograd.grad(d_loss_sr, [D[k] for k in dn])
    with torch.no_grad():
        adam_step(D, {k: a + b for k, a, b in zip(dn, g_hr, g_sr)}, d_opt, d_lr, betas, eps)
    for k in dn:
        D[k] = D[k].detach()
    # ---- G step ----
    pixel = pixel_weight * l1_mean(sr, gt)
    content = content_fn(sr.detach(), gt) if content_fn is not None else torch.zeros(1, 5)
    content = (content_weight * content).sum()
    adv_out = discriminator_unet_forward(sr, D, training=True)
    adv = adversarial_weight * bce_with_logits_mean(adv_out, 1.0)
    g_loss = pixel + content + adv
    if train_generator:
        grads = torch.autograd.grad(g_loss, [G[k] for k in gn])
        with torch.no_grad():
            adam_step(G, dict(zip(gn, grads)), g_opt, g_lr, betas, eps)
    for k in gn:
        G[k] = G[k].detach()
    return {
        "d_loss": float(d_loss_hr + d_loss_sr), "pixel_loss": float(pixel), "content_loss": float(content),
        "adversarial_loss": float(adv),
        "d_gt_probability": float(torch.sigmoid(gt_out.detach()).mean()),
        "d_sr_probability": float(torch.sigmoid(sr_out.detach()).mean()),
    }


def realesrgan_gan_step(G: Params, D: Params, g_opt: AdamState, d_opt: AdamState, lr_img: Tensor, gt: Tensor, gt_usm: Tensor, *,
                        upscale: int = 4, lr: float = 1e-4, betas: Tuple[float, float] = (0.9, 0.99), eps: float = 1e-4,
                        pixel_weight: float = 1.0, content_weight=(0.1, 0.1, 1.0, 1.0, 1.0), adversarial_weight: float = 0.1,
                        content_fn=None) -> Dict[str, float]:
    """One iteration of Real_ESRGAN/train_realesrgan.py:407-476 (realesrgan_config.py:138-151): GENERATOR first -- pixel and
    (detached, logged-only) content loss against the USM-sharpened GT, adversarial BCE vs ones through the frozen D -- Adam
    step; then D(gt) and D(sr.detach()) forward+backward (gradients accumulate), Adam step.  Probabilities are
    sigmoid(mean(logits)) (:475-476)."""
    gn, dn = g_param_names(G), d_param_names(D)
    _leafify(G, gn)
    for k in dn:
        D[k] = D[k].detach()
    sr = rrdbnet_forward(lr_img, G, upscale)
    pixel = pixel_weight * l1_mean(sr, gt_usm)
    content = content_fn(sr.detach(), gt_usm) if content_fn is not None else torch.zeros(1, 5)
    content = (torch.tensor(content_weight) * content).sum()
    adv = adversarial_weight * bce_with_logits_mean(discriminator_unet_forward(sr, D, training=True), 1.0)
    grads = torch.autograd.grad(pixel + content + adv, [G[k] for k in gn])
    with torch.no_grad():
        adam_step(G, dict(zip(gn, grads)), g_opt, lr, betas, eps)
    for k in gn:
        G[k] = G[k].detach()
    _leafify(D, dn)
    gt_out = discriminator_unet_forward(gt, D, training=True)
    d_loss_gt = bce_with_logits_mean(gt_out, 1.0)
    g1 = torch.autograd.grad(d_loss_gt, [D[k] for k in dn])
    sr_out = discriminator_unet_forward(sr.detach().clone(), D, training=True)
    d_loss_sr = bce_with_logits_mean(sr_out, 0.0)
    g2 = torch.autograd.grad(d_loss_sr, [D[k] for k in dn])
    with torch.no_grad():
        adam_step(D, {k: a + b for k, a, b in zip(dn, g1, g2)}, d_opt, lr, betas, eps)
    for k in dn:
        D[k] = D[k].detach()
    return {"d_loss": float(d_loss_gt + d_loss_sr), "pixel_loss": float(pixel), "content_loss": float(content), "adversarial_loss": float(adv),
            "d_gt_probability": float(torch.sigmoid(gt_out.detach().mean())), "d_sr_probability": float(torch.sigmoid(sr_out.detach().mean())),
            "sr": sr.detach()}


# ----------------------------------------------------------------------------------------------
# A-ESRGAN attention U-Net discriminator (BASELINE config 5)
# ----------------------------------------------------------------------------------------------
AESRGAN_SN_LAYERS = ("conv1", "conv2", "conv3", "gating", "cat_1.convU", "conv4", "cat_2.convU", "conv5",
                     "cat_3.convU", "conv6", "conv7", "conv8")


def aesrgan_unet_forward(x: Tensor, P: Params, training: bool = True, update_state: bool = True,
                         return_attention: bool = False):
    """UNetDiscriminatorAesrgan.forward -- A-ESRGAN/model.py:311-338 with add_attn.forward :239-254 and
    unetCat.forward :265-275.  P uses the reference's state_dict keys (``conv1.weight_orig/_u/_v``,
    ``attn_1.W.0.weight``, ``attn_1.W.1.running_mean`` ...).  In training mode the spectral-norm u/v and the
    BatchNorm running statistics in P are advanced exactly like the reference modules do."""
    def sn_w(layer: str) -> Tensor:
        w, u, v = spectral_norm_weight(P[f"{layer}.weight_orig"], P[f"{layer}.weight_u"], P[f"{layer}.weight_v"], training)
        if training and update_state:
            P[f"{layer}.weight_u"], P[f"{layer}.weight_v"] = u, v
        return w

    def lrelu(t: Tensor) -> Tensor:
        return F.leaky_relu(t, LRELU_SLOPE)

    def attn(xf: Tensor, g: Tensor, pre: str):
        theta = F.conv2d(xf, P[f"{pre}.theta.weight"], None, stride=2)
        phi = F.conv2d(g, P[f"{pre}.phi.weight"], P[f"{pre}.phi.bias"])
        phi = F.interpolate(phi, size=theta.shape[2:], mode="bilinear", align_corners=False)
        f = F.relu(theta + phi)
        sig = torch.sigmoid(F.conv2d(f, P[f"{pre}.psi.weight"], P[f"{pre}.psi.bias"]))
        sig = F.interpolate(sig, size=xf.shape[2:], mode="bilinear", align_corners=False)
        y = sig.expand_as(xf) * xf
        wy = F.conv2d(y, P[f"{pre}.W.0.weight"], P[f"{pre}.W.0.bias"])
        rm, rv = P[f"{pre}.W.1.running_mean"], P[f"{pre}.W.1.running_var"]
        if training and update_state:
            rm, rv = rm.clone(), rv.clone()
        wy = F.batch_norm(wy, rm if (training and update_state) or not training else None,
                          rv if (training and update_state) or not training else None,
                          P[f"{pre}.W.1.weight"], P[f"{pre}.W.1.bias"], training, 0.1, 1e-5)
        if training and update_state:
            P[f"{pre}.W.1.running_mean"], P[f"{pre}.W.1.running_var"] = rm, rv
            P[f"{pre}.W.1.num_batches_tracked"] = P[f"{pre}.W.1.num_batches_tracked"] + 1
        return wy, sig

    def cat(in1: Tensor, in2: Tensor, pre: str) -> Tensor:
        up = F.interpolate(in2, scale_factor=2, mode="bilinear", align_corners=False)
        out2 = lrelu(F.conv2d(up, sn_w(f"{pre}.convU"), None, padding=1))
        off = out2.shape[2] - in1.shape[2]
        out1 = F.pad(in1, 2 * [off // 2, off // 2])
        return torch.cat([out1, out2], 1)

    x0 = lrelu(F.conv2d(x, P["conv0.weight"], P["conv0.bias"], padding=1))
    x1 = lrelu(F.conv2d(x0, sn_w("conv1"), None, stride=2, padding=1))
    x2 = lrelu(F.conv2d(x1, sn_w("conv2"), None, stride=2, padding=1))
    x3 = lrelu(F.conv2d(x2, sn_w("conv3"), None, stride=2, padding=1))
    gated = lrelu(F.conv2d(x3, sn_w("gating"), None, padding=1))
    a1, s1 = attn(x2, gated, "attn_1")
    a2, s2 = attn(x1, gated, "attn_2")
    a3, s3 = attn(x0, gated, "attn_3")
    t = cat(a1, x3, "cat_1")
    x4 = lrelu(F.conv2d(t, sn_w("conv4"), None, padding=1))
    t = cat(a2, x4, "cat_2")
    x5 = lrelu(F.conv2d(t, sn_w("conv5"), None, padding=1))
    t = cat(a3, x5, "cat_3")
    x6 = lrelu(F.conv2d(t, sn_w("conv6"), None, padding=1))
    out = lrelu(F.conv2d(x6, sn_w("conv7"), None, padding=1))
    out = lrelu(F.conv2d(out, sn_w("conv8"), None, padding=1))
    out = F.conv2d(out, P["conv9.weight"], P["conv9.bias"], padding=1)
    return (out, (s1, s2, s3)) if return_attention else out


# ----------------------------------------------------------------------------------------------
# ESRGAN discriminator
# ----------------------------------------------------------------------------------------------
ESRGAN_D_CONVS = (0, 2, 5, 8, 11, 14, 17, 20, 23, 26)      # indices of the convs inside `features`
ESRGAN_D_STRIDES = (1, 2, 1, 2, 1, 2, 1, 2, 1, 2)


def esrgan_discriminator_forward(x: Tensor, P: Params, training: bool = True, update_state: bool = True,
                                 momentum: float = 0.1, eps: float = 1e-5) -> Tensor:
    """Discriminator.forward -- ESRGAN/model.py:88-141: conv3x3(bias)+LeakyReLU, then nine (conv, BatchNorm2d,
    LeakyReLU(0.2)) stages alternating 4x4 stride 2 / 3x3 stride 1, flatten (NCHW order), Linear(8192,100), LeakyReLU,
    Linear(100,1).  Training mode normalises with batch statistics and advances running_mean / running_var (unbiased
    variance, momentum 0.1) and num_batches_tracked in P, like nn.BatchNorm2d."""
    out = x
    for i, (fi, st) in enumerate(zip(ESRGAN_D_CONVS, ESRGAN_D_STRIDES)):
        w = P[f"features.{fi}.weight"]
        out = F.conv2d(out, w, P.get(f"features.{fi}.bias"), stride=st, padding=1)
        if i > 0:
            b = f"features.{fi + 1}"
            if training:
                mean = out.mean(dim=(0, 2, 3))
                var = out.var(dim=(0, 2, 3), unbiased=False)
                if update_state:
                    n = out.numel() / out.shape[1]
                    with torch.no_grad():
                        P[b + ".running_mean"] = (1 - momentum) * P[b + ".running_mean"] + momentum * mean.detach()
                        P[b + ".running_var"] = (1 - momentum) * P[b + ".running_var"] + momentum * var.detach() * n / (n - 1)
                        P[b + ".num_batches_tracked"] = P[b + ".num_batches_tracked"] + 1
            else:
                mean, var = P[b + ".running_mean"], P[b + ".running_var"]
            out = (out - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + eps)
            out = out * P[b + ".weight"][None, :, None, None] + P[b + ".bias"][None, :, None, None]
        out = F.leaky_relu(out, 0.2)
    out = torch.flatten(out, 1)
    out = F.leaky_relu(F.linear(out, P["classifier.0.weight"], P["classifier.0.bias"]), 0.2)
    return F.linear(out, P["classifier.2.weight"], P["classifier.2.bias"])


def content_loss_single(sr: Tensor, gt: Tensor, P: Params, node: str, mean: Sequence[float], std: Sequence[float]) -> Tensor:
    """ESRGAN ContentLoss.forward -- ESRGAN/model.py:281-292: normalise, run vgg19.features up to ONE node, F.l1_loss of the
    two feature maps; differentiable w.r.t. sr.  The extractor graph is cut at the node (create_feature_extractor), so the
    ReLU that follows it in torchvision's Sequential is not part of it: a conv node is observed pre-ReLU."""
    last = int(node.split(".")[1])
    m = torch.tensor(mean, dtype=sr.dtype).view(1, 3, 1, 1)
    s = torch.tensor(std, dtype=sr.dtype).view(1, 3, 1, 1)

    def run(x: Tensor) -> Tensor:
        x = (x - m) / s
        for kind, idx, _ in vgg19_feature_layers():
            if idx > last:
                break
            if kind == "conv":
                x = F.conv2d(x, P[f"features.{idx}.weight"], P[f"features.{idx}.bias"], padding=1)
                if idx != last:
                    x = F.relu(x)
            elif kind == "pool":
                x = F.max_pool2d(x, 2, 2)
        return x
    return F.l1_loss(run(sr), run(gt))


def esrgan_gan_step(G: Params, D: Params, g_opt: AdamState, d_opt: AdamState, lr_img: Tensor, gt: Tensor, *, upscale: int = 4,
                    lr: float = 1e-4, betas: Tuple[float, float] = (0.9, 0.99), eps: float = 1e-8, pixel_weight: float = 0.01,
                    content_weight: float = 1.0, adversarial_weight: float = 0.005, content_fn=None) -> Dict[str, float]:
    """One iteration of ESRGAN/train_esrgan.py:364-431: GENERATOR first (D frozen; relativistic-average adversarial term
    built from D(gt.detach()) and D(sr), both in training mode so BatchNorm statistics advance), Adam step; then the
    discriminator: D(gt), D(sr.detach()), loss on gt_output - mean(sr_output) (backward with retain_graph), a THIRD forward
    D(sr.detach()), loss on sr_output - mean(gt_output) (backward), Adam step."""
    gn, dn = g_param_names(G), [k for k in D if k.endswith((".weight", ".bias"))]
    _leafify(G, gn)
    for k in dn:
        D[k] = D[k].detach()
    sr = rrdbnet_forward(lr_img, G, upscale)
    gt_output = esrgan_discriminator_forward(gt.detach().clone(), D, training=True)
    sr_output = esrgan_discriminator_forward(sr, D, training=True)
    pixel = pixel_weight * l1_mean(sr, gt)
    content = content_weight * (content_fn(sr, gt) if content_fn is not None else torch.zeros(()))
    bce = F.binary_cross_entropy_with_logits
    adv = adversarial_weight * (bce(gt_output - sr_output.mean(), torch.zeros_like(gt_output)) * 0.5 +
                                bce(sr_output - gt_output.mean(), torch.ones_like(sr_output)) * 0.5)
    grads = torch.autograd.grad(pixel + content + adv, [G[k] for k in gn])
    with torch.no_grad():
        adam_step(G, dict(zip(gn, grads)), g_opt, lr, betas, eps)
    for k in gn:
        G[k] = G[k].detach()
    _leafify(D, dn)
    srd = sr.detach().clone()
    gt_output = esrgan_discriminator_forward(gt, D, training=True)
    sr_output = esrgan_discriminator_forward(srd, D, training=True)
    d_loss_gt = bce(gt_output - sr_output.mean(), torch.ones_like(gt_output)) * 0.5
    g1 = torch.autograd.grad(d_loss_gt, [D[k] for k in dn], retain_graph=True)
    sr_output = esrgan_discriminator_forward(srd, D, training=True)
    d_loss_sr = bce(sr_output - gt_output.mean(), torch.zeros_like(sr_output)) * 0.5
    g2 = torch.autograd.grad(d_loss_sr, [D[k] for k in dn])
    with torch.no_grad():
        adam_step(D, {k: a + b for k, a, b in zip(dn, g1, g2)}, d_opt, lr, betas, eps)
    for k in dn:
        D[k] = D[k].detach()
    return {"d_loss": float(d_loss_gt + d_loss_sr), "pixel_loss": float(pixel), "adversarial_loss": float(adv),
            "d_gt_probability": float(torch.sigmoid(gt_output.detach().mean())),
            "d_sr_probability": float(torch.sigmoid(sr_output.detach().mean())), "sr": sr.detach()}
