"""ctypes binding of libsrganfd_hip.so (include/srganfd.h).

The library is the product: there is no CPU or PyTorch fallback.  ``lib()`` raises if the shared
object has not been built (``python -c 'import __graft_entry__ as g; g.build()'``).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SRGANFD_LIB") or os.path.join(_HERE, "libsrganfd_hip.so")   # SRGANFD_LIB: A/B kernel builds (tools/)

HEADER = os.path.join(os.path.dirname(_HERE), "include", "srganfd.h")


_ABI_VERSION_BUILT = 7     # SRGANFD_ABI_VERSION this binding's structures were written against (tests/test_host_logic.py keeps it equal to the header's)


def _header_abi_version() -> int:
    """SRGANFD_ABI_VERSION of include/srganfd.h: the one constant the library and this binding share.  A copy of the package without
    the repository's include/ directory (an install, a vendored sub-tree) falls back to the value the binding was written against."""
    import re
    try:
        with open(HEADER) as f:
            m = re.search(r"^#define\s+SRGANFD_ABI_VERSION\s+(\d+)", f.read(), re.M)
    except OSError:
        return _ABI_VERSION_BUILT
    if not m:
        raise RuntimeError(f"{HEADER}: SRGANFD_ABI_VERSION not found")
    return int(m.group(1))


ABI_VERSION = _header_abi_version()
BF16, F32, F16 = 0, 1, 2
DT_NAME = {BF16: "bf16", F32: "f32", F16: "f16"}
ACT_NONE, ACT_LRELU, ACT_RELU = 0, 1, 2


class View(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("cstride", C.c_int32), ("c0", C.c_int32), ("planar", C.c_int32), ("pad_", C.c_int32)]


class ConvArgs(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("n", C.c_int32), ("h_in", C.c_int32), ("w_in", C.c_int32), ("up", C.c_int32),
        ("ksize", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32),
        ("cout_store", C.c_int32), ("h_out", C.c_int32), ("w_out", C.c_int32),
        ("x", View), ("y", View), ("r1", View), ("r2", View), ("mask", View),
        ("w_packed", C.c_void_p), ("bias", C.c_void_p), ("alpha_dev", C.c_void_p),
        ("alpha", C.c_float), ("slope", C.c_float), ("post_scale", C.c_float), ("r1_scale", C.c_float),
        ("r2_scale", C.c_float), ("mask_slope", C.c_float), ("act", C.c_int32), ("y_f32", C.c_int32),
        ("out_sy", C.c_int32), ("out_sx", C.c_int32), ("out_oy", C.c_int32), ("out_ox", C.c_int32),
        ("out_h_full", C.c_int32), ("out_w_full", C.c_int32), ("pad_y", C.c_int32), ("pad_x", C.c_int32),
        ("y2", View), ("out_classes", C.c_int32), ("class_pad_step", C.c_int32),
    ]


class ThinArgs(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("cs", C.c_int32), ("w_big_is_cout", C.c_int32),
        ("flip", C.c_int32), ("act", C.c_int32), ("slope", C.c_float), ("mask_slope", C.c_float),
        ("weight", C.c_void_p), ("bias", C.c_void_p), ("big", View), ("mask", View), ("thin", C.c_void_p), ("thin_out", C.c_void_p),
        ("thin_out_pitch", C.c_int32), ("pad_", C.c_int32),
    ]


class WgradReduceJob(C.Structure):
    _fields_ = [("plan_host", C.c_void_p), ("plan_dev", C.c_void_p), ("grads", C.c_void_p), ("scalars", C.c_void_p), ("workspace", C.c_void_p)]


class SnJob(C.Structure):
    _fields_ = [("w_orig", C.c_void_p), ("u", C.c_void_p), ("v", C.c_void_p), ("sigma_out", C.c_void_p), ("inv_sigma_out", C.c_void_p),
                ("workspace", C.c_void_p), ("rows", C.c_int32), ("cols", C.c_int32)]


class SnGradJob(C.Structure):
    _fields_ = [("g_weight", C.c_void_p), ("w_orig", C.c_void_p), ("u", C.c_void_p), ("v", C.c_void_p), ("inv_sigma", C.c_void_p),
                ("dw_orig", C.c_void_p), ("workspace", C.c_void_p), ("rows", C.c_int32), ("cols", C.c_int32)]


class PackSeg(C.Structure):
    _fields_ = [
        ("src_off", C.c_int64), ("scale_off", C.c_int64), ("co_src", C.c_int32), ("ci_src", C.c_int32),
        ("k_lo", C.c_int32), ("k_len", C.c_int32), ("co_off", C.c_int32), ("ci_off", C.c_int32),
        ("transposed", C.c_int32), ("scale", C.c_float),
    ]


class PackJob(C.Structure):
    _fields_ = [
        ("dst_off", C.c_int64), ("dtype", C.c_int32), ("ksize", C.c_int32), ("k", C.c_int32), ("n", C.c_int32),
        ("nseg", C.c_int32), ("layout", C.c_int32), ("seg", PackSeg * 5),
    ]


class WgradConv(C.Structure):
    _fields_ = [
        ("ci_lo", C.c_int32), ("cin", C.c_int32), ("co_lo", C.c_int32), ("cout", C.c_int32),
        ("dw_off", C.c_int64), ("db_off", C.c_int64), ("co_dst", C.c_int32), ("ci_dst", C.c_int32),
        ("alpha", C.c_float), ("beta", C.c_float), ("alpha_off", C.c_int64),
    ]


class WgradShape(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("n", C.c_int32), ("h_in", C.c_int32), ("w_in", C.c_int32), ("up", C.c_int32),
        ("ksize", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("h_out", C.c_int32), ("w_out", C.c_int32),
        ("x_channels", C.c_int32), ("dy_channels", C.c_int32), ("nconv", C.c_int32), ("splits", C.c_int32),
    ]


# symbol -> (restype, argtypes); tests check that the library exports every one of these
SYMBOLS = {
    "srganfd_last_error": (C.c_char_p, []),
    "srganfd_abi_version": (C.c_int, []),
    "srganfd_set_dry_run": (None, [C.c_int]),
    "srganfd_get_mfma16": (C.c_int, []),
    "srganfd_pack_layout": (C.c_int, [C.c_int32, C.c_int32, C.c_int32]),
    "srganfd_conv2d": (C.c_int, [C.POINTER(ConvArgs), C.c_void_p]),
    "srganfd_conv2d_describe": (C.c_int, [C.POINTER(ConvArgs), C.c_char_p, C.c_size_t]),
    "srganfd_dense_chain": (C.c_int, [C.POINTER(ConvArgs), C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]),
    "srganfd_dense_chain_check": (C.c_int, [C.POINTER(ConvArgs), C.c_int32]),
    "srganfd_dense_chain_workspace_bytes": (C.c_size_t, []),
    "srganfd_conv2d_thin_in": (C.c_int, [C.POINTER(ThinArgs), C.c_void_p]),
    "srganfd_conv2d_thin_out": (C.c_int, [C.POINTER(ThinArgs), C.c_void_p]),
    "srganfd_conv2d_thin_wgrad_workspace": (C.c_size_t, []),
    "srganfd_conv2d_thin_wgrad": (C.c_int, [C.POINTER(ThinArgs), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "srganfd_packed_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "srganfd_pack_weights": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "srganfd_wgrad_plan_bytes": (C.c_size_t, [C.POINTER(WgradShape), C.POINTER(WgradConv)]),
    "srganfd_wgrad_plan_build": (C.c_int, [C.POINTER(WgradShape), C.POINTER(WgradConv), C.c_void_p, C.c_size_t,
                                           C.POINTER(C.c_size_t)]),
    "srganfd_conv2d_wgrad_partial": (C.c_int, [C.c_void_p, C.c_void_p, View, View, C.c_void_p, C.c_size_t, C.c_void_p]),
    "srganfd_wgrad_reduce_batch": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "srganfd_conv2d_wgrad": (C.c_int, [C.c_void_p, C.c_void_p, View, View, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_size_t, C.c_void_p]),
    "srganfd_nchw_to_nhwc": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, View, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "srganfd_lrelu_bwd": (C.c_int, [View, View, View, View, C.c_int32, C.c_int64, C.c_int32, C.c_float, C.c_void_p]),
    "srganfd_nhwc_to_nchw": (C.c_int, [View, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "srganfd_clamp_grad_to_nhwc": (C.c_int, [C.c_void_p, View, C.c_int32, C.c_int32, C.c_int32, C.c_int32, View, C.c_int32, C.c_int32, C.c_void_p]),
    "srganfd_resample_bwd_lrelu": (C.c_int, [View, View, View, View, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p]),
    "srganfd_resample": (C.c_int, [C.c_int32, View, View, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "srganfd_axpby": (C.c_int, [View, View, C.c_int32, C.c_int64, C.c_int32, C.c_float, C.c_float, C.c_void_p]),
    "srganfd_l1_loss": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_int32, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "srganfd_l1_loss_views": (C.c_int, [View, View, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "srganfd_sigmoid_of_mean": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "srganfd_bce_logits": (C.c_int, [C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "srganfd_bce_logits_relativistic": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                                  C.c_void_p, C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "srganfd_spectral_norm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "srganfd_spectral_norm_grad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]),
    "srganfd_spectral_norm_batch": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_void_p]),
    "srganfd_spectral_norm_grad_batch": (C.c_int, [C.c_void_p, C.c_int32, C.c_float, C.c_void_p]),
    "srganfd_adam_ema": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float,
                                   C.c_float, C.c_int32, C.c_float, C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "srganfd_loss_scale_update": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int32, C.c_void_p]),
    "srganfd_nonfinite_flag": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p]),
    "srganfd_resize_bilinear": (C.c_int, [C.c_int32, View, View, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "srganfd_add_relu": (C.c_int, [View, View, View, C.c_int32, C.c_int64, C.c_int32, C.c_void_p]),
    "srganfd_sigmoid": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    "srganfd_sigmoid_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "srganfd_adam_ema_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float,
                                       C.c_float, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "srganfd_l1_grad_views": (C.c_int, [View, View, View, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_float, C.c_void_p]),
    "srganfd_maxpool2_relu_bwd": (C.c_int, [View, View, View, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "srganfd_nhwc_to_nchw_scaled": (C.c_int, [View, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "srganfd_crop_nchw": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int32] * 8 + [C.c_void_p]),
    "srganfd_u8hwc_to_nchw": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int32] * 8 + [C.c_float, C.c_void_p]),
    "srganfd_psnr": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int32] * 6 + [C.c_void_p, C.c_void_p, C.c_void_p]),
    "srganfd_filter2d": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int32] * 6 + [C.c_void_p, C.c_void_p]),
    "srganfd_filter2d_separable": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int32] * 6 + [C.c_void_p, C.c_void_p]),
    "srganfd_usm_sharp": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int32] * 6 + [C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "srganfd_diff_jpeg_table_floats": (C.c_int32, []),
    "srganfd_diff_jpeg_tables": (C.c_int, [C.c_void_p]),
    "srganfd_diff_jpeg": (C.c_int, [C.c_void_p] + [C.c_int32] * 4 + [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "srganfd_resize": (C.c_int, [C.c_void_p] + [C.c_int32] * 6 + [C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "srganfd_gaussian_noise": (C.c_int, [C.c_void_p] * 5 + [C.c_int32] * 6 + [C.c_void_p, C.c_void_p]),
    "srganfd_poisson_prepare": (C.c_int, [C.c_void_p] + [C.c_int32] * 5 + [C.c_void_p] * 6),
    "srganfd_poisson_apply": (C.c_int, [C.c_void_p] * 9 + [C.c_int32] * 6 + [C.c_void_p, C.c_void_p]),
    "srganfd_crop_rot_flip": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int32] * 8 + [C.c_void_p]),
    "srganfd_quantize_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "srganfd_ssim_workspace_doubles": (C.c_int64, [C.c_int32] * 7),
    "srganfd_ssim": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int32] * 6 + [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "srganfd_gate_mul": (C.c_int, [C.c_int32, View, C.c_void_p, View, View, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p]),
    "srganfd_batchnorm_fwd": (C.c_int, [View, View, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                        C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "srganfd_batchnorm_bwd": (C.c_int, [View, View, View, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                        C.c_void_p, C.c_void_p]),
    "srganfd_batchnorm_partial_floats": (C.c_int64, [C.c_int32]),
    "srganfd_batchnorm_fwd_sync": (C.c_int, [View, View, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                             C.c_float, C.c_void_p, C.c_void_p, C.c_float, C.c_int32, C.c_int64, C.c_void_p]),
    "srganfd_batchnorm_bwd_sync": (C.c_int, [View, View, View, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                             C.c_void_p, C.c_void_p, View, C.c_float, C.c_int32, C.c_int64, C.c_void_p]),
    "srganfd_batchnorm_act_fwd": (C.c_int, [View, View, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                            C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]),
    "srganfd_batchnorm_act_bwd": (C.c_int, [View, View, View, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                            C.c_void_p, View, C.c_float, C.c_void_p]),
}

LOSS_WS_FLOATS = 2049
SN_WS_FLOATS = 16 * 8192 + 1024        # srganfd_spectral_norm: ceil(rows / 32) * cols + rows for rows <= 512, cols <= 8192
SN_GRAD_WS_FLOATS = 1028               # srganfd_spectral_norm_grad: 1025, kept 16-byte aligned per job


def sn_ws_floats(rows: int, cols: int) -> int:
    """workspace of one srganfd_spectral_norm job (16-byte aligned)"""
    return ((rows + 31) // 32 * cols + rows + 3) // 4 * 4


_lib = None


class SrganfdError(RuntimeError):
    pass


def lib():
    """Load libsrganfd_hip.so (after torch, so both share torch's libamdhip64.so.7)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SrganfdError(
                f"{LIB_PATH} is missing: the HIP library is the product and has no fallback. "
                "Build it with `python -c 'import __graft_entry__ as g; g.build()'`.")
        import torch  # noqa: F401  (loads the HIP runtime the extension must share)
        l = C.CDLL(LIB_PATH)
        l.srganfd_abi_version.restype, l.srganfd_abi_version.argtypes = C.c_int, []
        have = l.srganfd_abi_version()
        if have != ABI_VERSION:
            raise SrganfdError(f"{LIB_PATH} implements ABI version {have}, this binding (include/srganfd.h) version {ABI_VERSION}: "
                               "argument lists differ -- rebuild it (`python __graft_entry__.py --force`, or tools/build_variant.sh for an "
                               "SRGANFD_LIB variant)")
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        for name in ("srganfd_set_debug", "srganfd_set_ring_mode", "srganfd_set_mfma16"):
            if hasattr(l, name):                   # -DSRGANFD_EXPERIMENT builds only (tools/build_variant.sh; select with SRGANFD_LIB)
                getattr(l, name).restype, getattr(l, name).argtypes = None, [C.c_int]
        _lib = l
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise SrganfdError(f"{what} failed (rc={rc}): {lib().srganfd_last_error().decode()}")


DRY_RUN = False


def set_dry_run(on: bool) -> None:
    """Host-logic tests on CPU: every entry point validates its arguments and launches nothing."""
    global DRY_RUN
    DRY_RUN = bool(on)
    lib().srganfd_set_dry_run(1 if on else 0)


def stream_ptr() -> int:
    if DRY_RUN:
        return 0
    import torch
    return torch.cuda.current_stream().cuda_stream


def view(t, cstride=None, c0=0, planar=0) -> View:
    """NHWC tensor (..., C) -> channel-slice view starting at channel c0 (planar=1: the buffer stores 32-channel group planes)."""
    return View(t.data_ptr(), int(cstride if cstride is not None else t.shape[-1]), int(c0), int(planar), 0)


NULL_VIEW = View(None, 0, 0, 0, 0)
