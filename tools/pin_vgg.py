#!/usr/bin/env python3
"""Pin the VGG-19 content loss against torchvision -- for a machine that HAS torchvision and the ImageNet weights.

The build container has neither (no network, no ~/.cache/torch), so the content-loss values of this package are checked against
its own CPU restatement only ("parity unpinned").  Run this script wherever both exist:

    python tools/pin_vgg.py [--reference /path/to/SR-GAN-FD] [--weights vgg19-dcbb9e9d.pth] [--out tests/golden/vgg19_taps.npz]

It evaluates the reference's ``ContentLoss`` on seeded inputs and writes inputs + expected values as a small fixture:
  * with ``--reference`` the reference's own class is imported (BSRGAN/model.py:501-554: ``models.vgg19(IMAGENET1K_V1)``,
    ``create_feature_extractor(model, nodes)``, ``transforms.Normalize``, ``torch.Tensor([losses])``) and called as the train script
    calls it (train_bsrgan.py:298-300,453) -- this is the real pin;
  * without it the same four statements are issued on torchvision directly (identical semantics; the fixture records which was used).
ESRGAN's single-node form (ESRGAN/model.py:258-292, node ``features.34``) is recorded the same way.

The fixture holds DATA only: the two input batches, the (1, 5) BSRGAN values, the ESRGAN scalar, per-node mean / mean-abs of the SR
taps (to localise a mismatch) and the SHA-256 of the features' weights.  The weights themselves (80 MB) are not stored:
``tests/test_gan_gpu.py::test_content_loss_vs_pinned_torchvision_taps`` runs when the fixture exists AND the environment variable
``SRGANFD_VGG19_WEIGHTS`` names the same torchvision ``vgg19`` state_dict (checked by that hash), and compares the HIP path's
``ContentLoss(nodes, mean, std, weights_path=...)`` with the recorded values at 1e-3 (f32) / 2e-3 (f16).
"""
from __future__ import annotations

import argparse
import hashlib
import os
import sys

import numpy as np
import torch

NODES = ["features.2", "features.7", "features.16", "features.25", "features.34"]      # bsrgan_config.py:130
MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]                                 # bsrgan_config.py:131-132


def features_sha256(state_dict) -> str:
    """hash of the ``features.*`` tensors in key order (what both the fixture and the test compute from a vgg19 state_dict)"""
    h = hashlib.sha256()
    for k in sorted(state_dict):
        if k.startswith("features."):
            h.update(k.encode())
            h.update(state_dict[k].detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def main() -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--reference", default="", help="checkout of MiNeves00/SR-GAN-FD: import its BSRGAN/model.py and ESRGAN/model.py ContentLoss")
    ap.add_argument("--weights", default="", help="torchvision vgg19 state_dict file (default: let torchvision download / use its cache)")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "vgg19_taps.npz"))
    ap.add_argument("--size", type=int, nargs=2, default=[64, 96], help="H W of the seeded inputs (multiples of 16)")
    a = ap.parse_args()
    try:
        import torchvision
        from torchvision import models, transforms
        from torchvision.models.feature_extraction import create_feature_extractor
    except ImportError as e:
        print(f"pin_vgg: torchvision is not installed here ({e}); run this on a machine that has it -- nothing written", file=sys.stderr)
        return 2

    if a.weights:
        # the reference calls models.vgg19(weights=IMAGENET1K_V1): serve that call from the given file instead of the network
        sd = torch.load(a.weights, map_location="cpu")
        _orig = models.vgg19

        def vgg19_from_file(*args, **kw):
            kw.pop("weights", None)
            m = _orig(weights=None)
            m.load_state_dict(sd)
            return m
        models.vgg19 = vgg19_from_file
    torch.manual_seed(3)
    h, w = a.size
    sr, gt = torch.rand(2, 3, h, w), torch.rand(2, 3, h, w)
    out = {"sr": sr.numpy(), "gt": gt.numpy(), "nodes": np.array(NODES), "mean": np.array(MEAN), "std": np.array(STD),
           "torchvision_version": np.array(torchvision.__version__), "torch_version": np.array(torch.__version__)}

    if a.reference:
        # the reference's own classes, called as its train scripts call them
        import importlib.util

        def load(path, name):
            sys.path.insert(0, os.path.dirname(path))
            spec = importlib.util.spec_from_file_location(name, path)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            sys.path.pop(0)
            return mod
        bs = load(os.path.join(a.reference, "BSRGAN", "model.py"), "ref_bsrgan_model")
        cl5 = bs.ContentLoss(NODES, MEAN, STD).eval()
        with torch.no_grad():
            v5 = cl5(sr, gt)
        fx = cl5.feature_extractor
        es = load(os.path.join(a.reference, "ESRGAN", "model.py"), "ref_esrgan_model")
        cl1 = es.content_loss(feature_model_extractor_node="features.34", feature_model_normalize_mean=MEAN, feature_model_normalize_std=STD).eval()
        s1 = sr.clone().requires_grad_(True)
        v1 = cl1(s1, gt)
        v1.backward()
        out["source"] = np.array("reference classes (BSRGAN/model.py:501-554, ESRGAN/model.py:258-292)")
        model_sd = fx.state_dict()
    else:
        model = models.vgg19(weights=models.VGG19_Weights.IMAGENET1K_V1)          # model.py:522
        fx = create_feature_extractor(model, NODES).eval()                        # :524-526
        norm = transforms.Normalize(MEAN, STD)                                    # :530
        with torch.no_grad():
            fs, fg = fx(norm(sr)), fx(norm(gt))                                   # :542-546
            v5 = torch.Tensor([[torch.nn.functional.l1_loss(fs[n], fg[n]) for n in NODES]])      # :548-552
        fx1 = create_feature_extractor(models.vgg19(weights=models.VGG19_Weights.IMAGENET1K_V1), ["features.34"]).eval()
        s1 = sr.clone().requires_grad_(True)
        v1 = torch.nn.functional.l1_loss(fx1(norm(s1))["features.34"], fx1(norm(gt))["features.34"])        # ESRGAN/model.py:281-292
        v1.backward()
        out["source"] = np.array("torchvision directly, statements of BSRGAN/model.py:522-552 and ESRGAN/model.py:281-292")
        model_sd = fx.state_dict()
    with torch.no_grad():
        taps = fx((sr - torch.tensor(MEAN).view(1, 3, 1, 1)) / torch.tensor(STD).view(1, 3, 1, 1))
    out["bsrgan_values"] = v5.detach().cpu().numpy().astype(np.float32)           # (1, 5)
    out["esrgan_value"] = np.float32(v1.item())
    out["esrgan_dsr"] = s1.grad.numpy()
    for n in NODES:
        out[f"tap_stats/{n}"] = np.array([taps[n].mean().item(), taps[n].abs().mean().item(), float((taps[n] < 0).any())], dtype=np.float64)
    out["features_sha256"] = np.array(features_sha256({k: v for k, v in model_sd.items()}))
    np.savez_compressed(a.out, **out)
    print(f"pin_vgg: wrote {a.out}: BSRGAN values {out['bsrgan_values']}, ESRGAN value {out['esrgan_value']:.6f}; "
          f"negative values in taps (1 = observed pre-ReLU): { {n: int(out[f'tap_stats/{n}'][2]) for n in NODES} }")
    return 0


if __name__ == "__main__":
    sys.exit(main())
