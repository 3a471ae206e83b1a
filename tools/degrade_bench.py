#!/usr/bin/env python3
"""Throughput of the on-device degradation stages at Real-ESRGAN's training shape (batch 48, 3x256x256 GT,
realesrgan_config.py:116-117), each stage against the roof that bounds it, plus the whole degradation_process (the CPU side of
the comparison is bench.py's cpu_baseline leg of `--workload realesrgan_gan`; the oracle is not used from tools/).

    python tools/degrade_bench.py [--batch 48] [--size 256] [--iters 20]

Prints one JSON line per stage: {"stage", "us", "GB/s" (algorithmic bytes: input read once + output written once),
"hbm_frac" (of 8 TB/s), "GFLOP/s", "valu_frac" (of 157.3 TFLOP/s fp32 vector)}.  Timed with HIP events on the current stream.
"""
import argparse
import json
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import importlib

PEAK_HBM, PEAK_VALU = 8000e9, 157.3e12
PARAMS = dict(first_blur_probability=1.0, resize_probability1=[0.2, 0.7, 0.1], resize_range1=[0.15, 1.5], gray_noise_probability1=0.4,
              gaussian_noise_probability1=0.5, noise_range1=[1, 30], poisson_scale_range1=[0.05, 3], jpeg_range1=[30, 95],
              second_blur_probability=0.8, resize_probability2=[0.3, 0.4, 0.3], resize_range2=[0.3, 1.2], gray_noise_probability2=0.4,
              gaussian_noise_probability2=0.5, noise_range2=[1, 25], poisson_scale_range2=[0.05, 2.5], jpeg_range2=[30, 95])


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=48)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    imgproc = importlib.import_module("sr_gan_fd_amd.imgproc")
    b, n = a.batch, a.size
    torch.manual_seed(0)
    gt = torch.rand(b, 3, n, n, device="cuda")
    k21 = torch.rand(b, 21, 21, device="cuda")
    k21 = k21 / k21.sum(dim=(1, 2), keepdim=True)
    usm, jpeg = imgproc.USMSharp().cuda(), imgproc.DiffJPEG().cuda()
    img_bytes = gt.numel() * 4
    px = b * 3 * n * n
    quality = torch.full((b,), 60.0, device="cuda")
    randn = torch.randn_like(gt)
    sigma, gray = torch.full((b,), 10.0, device="cuda"), torch.zeros(b, device="cuda")
    stages = [
        ("filter2d 21x21 per-image kernels", lambda: imgproc.filter2d_torch(gt, k21), 2 * img_bytes, 2.0 * px * 441),
        ("USMSharp 51x51 (two fused separable passes)", lambda: usm(gt, 0.5, 10), 7 * img_bytes, 2.0 * px * (51 * 82 / 32 + 51) * 2),
        ("DiffJPEG round trip", lambda: jpeg(gt, quality.clone()), 2 * img_bytes, 2.0 * px * 64 * 2 * 1.5),
        ("resize bicubic x0.5", lambda: imgproc.interpolate(gt, scale_factor=0.5, mode="bicubic"), 1.25 * img_bytes, 2.0 * px / 4 * 20),
        ("resize area -> 64x64", lambda: imgproc.interpolate(gt, size=(n // 4, n // 4), mode="area"), (1 + 1 / 16) * img_bytes, px),
        ("gaussian noise apply (draws given)", lambda: imgproc.gaussian_noise_apply(gt, randn, None, sigma, gray, True, False), 3 * img_bytes, 3.0 * px),
        ("quantize_u8", lambda: imgproc.quantize_u8(gt), 2 * img_bytes, 3.0 * px),
    ]
    for name, fn, nbytes, flop in stages:
        t = timed(fn, a.iters)
        print(json.dumps({"stage": name, "us": round(t * 1e6, 1), "GB/s": round(nbytes / t / 1e9, 1), "hbm_frac": round(nbytes / t / PEAK_HBM, 4),
                          "GFLOP/s": round(flop / t / 1e9, 1), "valu_frac": round(flop / t / PEAK_VALU, 4)}))

    def pipeline():
        return imgproc.degradation_process(gt, k21, k21, k21, 4, PARAMS, jpeg, usm)
    random.seed(0); np.random.seed(0)
    t = timed(pipeline, a.iters)
    line = {"stage": "degradation_process (USM + 2nd-order pipeline, random branches)", "ms": round(t * 1e3, 3), "img/s": round(b / t, 1),
            "batch": b, "gt": f"3x{n}x{n}"}
    print(json.dumps(line))


if __name__ == "__main__":
    main()
