"""Drop-in mirror of the reference's ``model.py`` module surface, running on hand-written HIP kernels.

Same names, constructor kwargs, ``state_dict`` keys/shapes (NCHW fp32 parameters) and ``nn.Module``
protocol as ``BSRGAN/model.py`` (BSRGAN :311-384, DiscriminatorUNet :91-167, ContentLoss :501-554,
factories :557-593) and ``ESRGAN/model.py`` (RRDBNet :144-243, factories :301-322), so that
``model.__dict__[cfg.g_model_arch_name](**kwargs)`` (train_bsrgan.py:274-285), ``AveragedModel``
deep copies, pickling and ``utils.load_state_dict`` keep working.

The sub-modules (``nn.Conv2d``, ``spectral_norm``) are used ONLY as parameter containers: they give
the reference's parameter names, shapes, construction order and therefore the same RNG stream at
init.  Their ``forward`` is never called -- ``forward`` here hands the parameters to the HIP engine
(``engine.py``), which raises if libsrganfd_hip.so or a GPU is missing (no CPU fallback).

Precision follows the reference's own contract (``engine.resolve_compute_dtype``): inside ``torch.autocast("cuda")`` -- the
train loops' ``amp.autocast()``, train_bsrgan.py:415-427,450-457 -- the kernels compute in the autocast dtype (float16 unless
the caller asked autocast for bfloat16: f16 MFMA, fp32 accumulate, fp32 master weights, fp32 SR / logits); outside it --
``validate()``, train_bsrgan.py:563, inference.py -- in float32 (exact-fp32 MFMA, the mode every golden vector is met in).
One knob beyond the reference: ``compute_dtype`` (default ``None`` = follow autocast) pins a module to torch.float32 /
float16 / bfloat16 whatever the autocast state; the fused trainers and bench.py set it explicitly.
"""
from __future__ import annotations

from typing import Any, List

import torch
from torch import Tensor, nn
from torch.nn.utils import spectral_norm

__all__ = [
    "DiscriminatorUNet", "BSRGAN", "RRDBNet", "ContentLoss",
    "discriminator_unet", "bsrgan_x2", "bsrgan_x4", "content_loss",
    "rrdbnet_x1", "rrdbnet_x2", "rrdbnet_x4", "rrdbnet_x8",
    "UNetDiscriminatorAesrgan", "uNetDiscriminatorAesrgan",
    "Discriminator", "discriminator",
]


class _ResidualDenseBlock(nn.Module):
    """Parameter container of BSRGAN/model.py:31-62; standalone forward = a 1-block trunk on the GPU."""

    def __init__(self, channels: int, growth_channels: int, self_init: bool = False) -> None:
        super().__init__()
        self.conv1 = nn.Conv2d(channels + growth_channels * 0, growth_channels, (3, 3), (1, 1), (1, 1))
        self.conv2 = nn.Conv2d(channels + growth_channels * 1, growth_channels, (3, 3), (1, 1), (1, 1))
        self.conv3 = nn.Conv2d(channels + growth_channels * 2, growth_channels, (3, 3), (1, 1), (1, 1))
        self.conv4 = nn.Conv2d(channels + growth_channels * 3, growth_channels, (3, 3), (1, 1), (1, 1))
        self.conv5 = nn.Conv2d(channels + growth_channels * 4, channels, (3, 3), (1, 1), (1, 1))
        self.leaky_relu = nn.LeakyReLU(0.2, True)
        self.identity = nn.Identity()
        if self_init:     # Real_ESRGAN/model.py:135-150: its dense block initialises itself (the generator then re-draws everything)
            for module in self.modules():
                if isinstance(module, nn.Conv2d):
                    nn.init.kaiming_normal_(module.weight)
                    module.weight.data *= 0.1
                    if module.bias is not None:
                        nn.init.constant_(module.bias, 0)
        self.compute_dtype = None

    def forward(self, x: Tensor) -> Tensor:
        from .engine import trunk_apply
        return trunk_apply(self, x, [self], rrdb=False)


class _ResidualResidualDenseBlock(nn.Module):
    """Parameter container of BSRGAN/model.py:65-88."""

    def __init__(self, channels: int, growth_channels: int, self_init: bool = False) -> None:
        super().__init__()
        self.rdb1 = _ResidualDenseBlock(channels, growth_channels, self_init)
        self.rdb2 = _ResidualDenseBlock(channels, growth_channels, self_init)
        self.rdb3 = _ResidualDenseBlock(channels, growth_channels, self_init)
        self.compute_dtype = None

    def forward(self, x: Tensor) -> Tensor:
        from .engine import trunk_apply
        return trunk_apply(self, x, [self.rdb1, self.rdb2, self.rdb3], rrdb=True)


class _RRDBGenerator(nn.Module):
    """Shared body of BSRGAN (BSRGAN/model.py:311-384) and RRDBNet (ESRGAN/model.py:144-243)."""

    def __init__(self, in_channels: int, out_channels: int, channels: int, growth_channels: int,
                 num_blocks: int, upscale_factor: int, always_up1: bool, rdb_self_init: bool = False, unshuffle: int = 1) -> None:
        super().__init__()
        self.upscale_factor = upscale_factor
        # Real-ESRGAN below x4 (Real_ESRGAN/model.py:190-204,248): the input is pixel-unshuffled by `unshuffle` (conv1 sees
        # in_channels * unshuffle^2 channels at 1 / unshuffle of the size) and BOTH nearest-x2 stages always run
        self.unshuffle = unshuffle
        if unshuffle > 1:
            in_channels *= unshuffle * unshuffle
            self.downsampling = nn.PixelUnshuffle(unshuffle)
        self.conv1 = nn.Conv2d(in_channels, channels, (3, 3), (1, 1), (1, 1))
        self.trunk = nn.Sequential(*[_ResidualResidualDenseBlock(channels, growth_channels, rdb_self_init) for _ in range(num_blocks)])
        self.conv2 = nn.Conv2d(channels, channels, (3, 3), (1, 1), (1, 1))
        n_up = {1: 0, 2: 1, 4: 2, 8: 3}[upscale_factor * unshuffle]
        if always_up1:               # BSRGAN builds upsampling1 unconditionally (model.py:337-340)
            n_up = max(n_up, 1)
        for u in range(1, n_up + 1):
            setattr(self, f"upsampling{u}", nn.Sequential(
                nn.Conv2d(channels, channels, (3, 3), (1, 1), (1, 1)), nn.LeakyReLU(0.2, True)))
        self.conv3 = nn.Sequential(nn.Conv2d(channels, channels, (3, 3), (1, 1), (1, 1)), nn.LeakyReLU(0.2, True))
        self.conv4 = nn.Conv2d(channels, out_channels, (3, 3), (1, 1), (1, 1))
        # BSRGAN/model.py:358-363, ESRGAN/model.py:236-242
        for module in self.modules():
            if isinstance(module, nn.Conv2d):
                nn.init.kaiming_normal_(module.weight)
                module.weight.data *= 0.1
                if module.bias is not None:
                    nn.init.constant_(module.bias, 0)
        self.compute_dtype = None

    def n_upsample(self) -> int:
        return {1: 0, 2: 1, 4: 2, 8: 3}[self.upscale_factor * getattr(self, "unshuffle", 1)]

    def forward(self, x: Tensor) -> Tensor:
        return self._forward_impl(x)

    def _forward_impl(self, x: Tensor) -> Tensor:
        from .engine import generator_apply
        if getattr(self, "unshuffle", 1) > 1:
            # a pure re-indexing of the input image (no parameters, no arithmetic; the input needs no gradient): torch's op, then the
            # engine's conv1 reads 12 / 48 channels (padded to 32 / 64 for the MFMA path)
            x = torch.nn.functional.pixel_unshuffle(x, self.unshuffle)
        return generator_apply(self, x)


class BSRGAN(_RRDBGenerator):
    def __init__(self, in_channels: int = 3, out_channels: int = 3, channels: int = 64, growth_channels: int = 32,
                 num_rrdb: int = 23, upscale_factor: int = 4) -> None:
        super().__init__(in_channels, out_channels, channels, growth_channels, num_rrdb, upscale_factor, always_up1=True)


class RRDBNet(_RRDBGenerator):
    """ESRGAN/model.py:144-243 (``num_blocks``) and Real_ESRGAN/model.py:179-262 (``num_rrdb``: x4 as its shipped factory --
    PixelUnshuffle is the identity --, x2 / x1 with PixelUnshuffle(2 / 4) in front of conv1; both upsampling stages always run)."""

    def __init__(self, in_channels: int = 3, out_channels: int = 3, channels: int = 64, growth_channels: int = 32,
                 num_blocks: int = 23, upscale_factor: int = 4, num_rrdb: int = None) -> None:
        unshuffle = 1
        if num_rrdb is not None:
            if upscale_factor not in (1, 2, 4):
                raise ValueError("Real-ESRGAN's RRDBNet takes upscale_factor 1, 2 or 4 (Real_ESRGAN/model.py:190-201)")
            unshuffle = 4 // upscale_factor       # x2: PixelUnshuffle(2), x1: PixelUnshuffle(4), x4: identity
            num_blocks = num_rrdb
        # Real-ESRGAN's dense blocks draw their own initialisation first: same random stream, so a seeded construction
        # gives the reference's initial weights
        super().__init__(in_channels, out_channels, channels, growth_channels, num_blocks, upscale_factor, always_up1=False,
                         rdb_self_init=num_rrdb is not None, unshuffle=unshuffle)


class DiscriminatorUNet(nn.Module):
    """BSRGAN/model.py:91-167 (identical Real_ESRGAN/model.py:29-105)."""

    def __init__(self, in_channels: int, out_channels: int, channels: int, upsample_method: str = "bilinear") -> None:
        super().__init__()
        # stored and never read, exactly as the reference does (BSRGAN/model.py:97-100): its forward hard-codes
        # F.interpolate(..., mode="bilinear") at :150,154,158 whatever this says, and so does the HIP path
        self.upsample_method = upsample_method
        self.conv1 = nn.Conv2d(in_channels, 64, (3, 3), (1, 1), (1, 1))

        def sn(cin, cout, k, s):
            return nn.Sequential(spectral_norm(nn.Conv2d(cin, cout, (k, k), (s, s), (1, 1), bias=False)),
                                 nn.LeakyReLU(0.2, True))
        self.down_block1 = sn(channels, int(channels * 2), 4, 2)
        self.down_block2 = sn(int(channels * 2), int(channels * 4), 4, 2)
        self.down_block3 = sn(int(channels * 4), int(channels * 8), 4, 2)
        self.up_block1 = sn(int(channels * 8), int(channels * 4), 3, 1)
        self.up_block2 = sn(int(channels * 4), int(channels * 2), 3, 1)
        self.up_block3 = sn(int(channels * 2), channels, 3, 1)
        self.conv2 = sn(channels, channels, 3, 1)
        self.conv3 = sn(channels, channels, 3, 1)
        self.conv4 = nn.Conv2d(channels, out_channels, (3, 3), (1, 1), (1, 1))
        self.compute_dtype = None

    def forward(self, x: Tensor) -> Tensor:
        return self._forward_impl(x)

    def _forward_impl(self, x: Tensor) -> Tensor:
        from .engine import discriminator_apply
        return discriminator_apply(self, x)


class ContentLoss(nn.Module):
    """BSRGAN/model.py:501-554.  torchvision and the ImageNet VGG-19 weights are not part of the
    reference tree (third-party, network download), so the VGG-19 ``features[0:35]`` topology is
    restated here and its weights come from ``weights_path`` (a torchvision ``vgg19`` state_dict) or,
    when absent, from a seeded random init (benchmarks).  The returned tensor is DETACHED with shape
    (1, len(nodes)) exactly like the reference's ``torch.Tensor([losses])`` (:552).

    ESRGAN's variant (ESRGAN/model.py:258-292) passes ONE node name as a string and keeps the result in the autograd
    graph (``F.l1_loss`` of the two feature maps, a scalar): given a ``str`` this class behaves that way, with the frozen
    extractor's backward pass on the HIP engine (engine_v.py)."""

    def __init__(self, feature_model_extractor_nodes, feature_model_normalize_mean: list,
                 feature_model_normalize_std: list, weights_path: str = "", taps_post_relu: bool = True) -> None:
        super().__init__()
        from .vgg import build_vgg19_features
        self.single_node = isinstance(feature_model_extractor_nodes, str)
        if self.single_node:
            feature_model_extractor_nodes = [feature_model_extractor_nodes]
        self.feature_model_extractor_nodes = list(feature_model_extractor_nodes)
        self.taps_post_relu = taps_post_relu
        self.features = build_vgg19_features(weights_path)
        self.register_buffer("mean", torch.tensor(feature_model_normalize_mean, dtype=torch.float32))
        self.register_buffer("std", torch.tensor(feature_model_normalize_std, dtype=torch.float32))
        for p in self.features.parameters():
            p.requires_grad = False
        self.compute_dtype = None

    def forward(self, sr_tensor: Tensor, gt_tensor: Tensor) -> Tensor:
        assert sr_tensor.size() == gt_tensor.size(), "Two tensor must have the same size"
        if self.single_node:
            from .engine_v import content_loss_single_apply
            return content_loss_single_apply(self, sr_tensor, gt_tensor)
        from .engine import content_loss_apply
        return content_loss_apply(self, sr_tensor, gt_tensor)


class add_attn(nn.Module):
    """Parameter container of the attention gate, A-ESRGAN/model.py:228-254."""

    def __init__(self, x_channels, g_channels=256):
        super().__init__()
        self.W = nn.Sequential(nn.Conv2d(x_channels, x_channels, kernel_size=1, stride=1, padding=0), nn.BatchNorm2d(x_channels))
        self.theta = nn.Conv2d(x_channels, x_channels, kernel_size=2, stride=2, padding=0, bias=False)
        self.phi = nn.Conv2d(g_channels, x_channels, kernel_size=1, stride=1, padding=0, bias=True)
        self.psi = nn.Conv2d(x_channels, out_channels=1, kernel_size=1, stride=1, padding=0, bias=True)


class unetCat(nn.Module):
    """Parameter container of A-ESRGAN/model.py:258-275."""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.convU = spectral_norm(nn.Conv2d(dim_in, dim_out, 3, 1, 1, bias=False))


class UNetDiscriminatorAesrgan(nn.Module):
    """A-ESRGAN/model.py:279-345: attention U-Net discriminator with spectral normalisation (BASELINE config 5).
    Same constructor arguments, state_dict keys and construction order (hence the same init RNG stream)."""

    def __init__(self, num_in_ch, num_feat=64, skip_connection=True):
        super().__init__()
        norm = spectral_norm
        self.conv0 = nn.Conv2d(num_in_ch, num_feat, kernel_size=3, stride=1, padding=1)
        self.conv1 = norm(nn.Conv2d(num_feat, num_feat * 2, 3, 2, 1, bias=False))
        self.conv2 = norm(nn.Conv2d(num_feat * 2, num_feat * 4, 3, 2, 1, bias=False))
        self.conv3 = norm(nn.Conv2d(num_feat * 4, num_feat * 8, 3, 2, 1, bias=False))
        self.gating = norm(nn.Conv2d(num_feat * 8, num_feat * 4, 1, 1, 1, bias=False))
        self.attn_1 = add_attn(x_channels=num_feat * 4, g_channels=num_feat * 4)
        self.attn_2 = add_attn(x_channels=num_feat * 2, g_channels=num_feat * 4)
        self.attn_3 = add_attn(x_channels=num_feat, g_channels=num_feat * 4)
        self.cat_1 = unetCat(dim_in=num_feat * 8, dim_out=num_feat * 4)
        self.cat_2 = unetCat(dim_in=num_feat * 4, dim_out=num_feat * 2)
        self.cat_3 = unetCat(dim_in=num_feat * 2, dim_out=num_feat)
        self.conv4 = norm(nn.Conv2d(num_feat * 8, num_feat * 4, 3, 1, 1, bias=False))
        self.conv5 = norm(nn.Conv2d(num_feat * 4, num_feat * 2, 3, 1, 1, bias=False))
        self.conv6 = norm(nn.Conv2d(num_feat * 2, num_feat, 3, 1, 1, bias=False))
        self.conv7 = norm(nn.Conv2d(num_feat, num_feat, 3, 1, 1, bias=False))
        self.conv8 = norm(nn.Conv2d(num_feat, num_feat, 3, 1, 1, bias=False))
        self.conv9 = nn.Conv2d(num_feat, 1, 3, 1, 1)
        self.compute_dtype = None
        self.ly1 = self.ly2 = self.ly3 = None

    def forward(self, x: Tensor) -> Tensor:
        from .engine_a import aesrgan_discriminator_apply
        return aesrgan_discriminator_apply(self, x)

    def getAttentionLayers(self):
        return self.ly1, self.ly2, self.ly3


class Discriminator(nn.Module):
    """ESRGAN/model.py:88-141: VGG-style conv / BatchNorm2d / LeakyReLU stack (3x128x128 -> 512x4x4) and a two-layer
    classifier; used by the relativistic GAN step of ESRGAN/train_esrgan.py:370-425.  Same ``features.*`` /
    ``classifier.*`` state_dict keys; the sub-modules are parameter containers, forward/backward run on the HIP engine."""

    def __init__(self) -> None:
        super().__init__()
        layers = [nn.Conv2d(3, 64, (3, 3), (1, 1), (1, 1), bias=True), nn.LeakyReLU(0.2, True)]
        cin = 64
        for cout, k, s in ((64, 4, 2), (128, 3, 1), (128, 4, 2), (256, 3, 1), (256, 4, 2), (512, 3, 1), (512, 4, 2), (512, 3, 1), (512, 4, 2)):
            layers += [nn.Conv2d(cin, cout, (k, k), (s, s), (1, 1), bias=False), nn.BatchNorm2d(cout), nn.LeakyReLU(0.2, True)]
            cin = cout
        self.features = nn.Sequential(*layers)
        self.classifier = nn.Sequential(nn.Linear(512 * 4 * 4, 100), nn.LeakyReLU(0.2, True), nn.Linear(100, 1))
        self.compute_dtype = None

    def forward(self, x: Tensor) -> Tensor:
        from .engine_e import esrgan_discriminator_apply
        return esrgan_discriminator_apply(self, x)


def discriminator() -> Discriminator:
    """ESRGAN/model.py:293-296"""
    return Discriminator()


def uNetDiscriminatorAesrgan() -> UNetDiscriminatorAesrgan:
    return UNetDiscriminatorAesrgan(3)


def discriminator_unet(**kwargs: Any) -> DiscriminatorUNet:
    return DiscriminatorUNet(**kwargs)


def bsrgan_x2(**kwargs: Any) -> BSRGAN:
    return BSRGAN(upscale_factor=2, **kwargs)


def bsrgan_x4(**kwargs: Any) -> BSRGAN:
    return BSRGAN(upscale_factor=4, **kwargs)


def content_loss(*args: Any, **kwargs: Any) -> ContentLoss:
    """BSRGAN/model.py:568-571 (keyword form) and ESRGAN/model.py:325-332, which passes (node, mean, std) positionally"""
    return ContentLoss(*args, **kwargs)


def rrdbnet_x1(**kwargs: Any) -> RRDBNet:
    return RRDBNet(upscale_factor=1, **kwargs)


def rrdbnet_x2(**kwargs: Any) -> RRDBNet:
    return RRDBNet(upscale_factor=2, **kwargs)


def rrdbnet_x4(**kwargs: Any) -> RRDBNet:
    return RRDBNet(upscale_factor=4, **kwargs)


def rrdbnet_x8(**kwargs: Any) -> RRDBNet:
    return RRDBNet(upscale_factor=8, **kwargs)
