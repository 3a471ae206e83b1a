"""Shared helpers for the test-suite (golden loading, checksums, init recipe)."""
import os

import numpy as np
import torch


def load_golden(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def table(g, key):
    """checksum table saved by make_golden.save(): (n,3) array + key list -> dict"""
    return {str(k): v for k, v in zip(g[key + "__keys"], g[key])}


def checksum(t):
    a = torch.as_tensor(t).detach().double().flatten().cpu()
    w = torch.cos(torch.arange(a.numel(), dtype=torch.float64) * 0.37)
    return np.array([a.sum().item(), a.abs().sum().item(), (a * w).sum().item()])


def scaled_init(g, scale, bias):
    """SURVEY 8c init recipe (default init is degenerate: SR std 5e-5, 35 % of pixels clamped)."""
    with torch.no_grad():
        for p in g.parameters():
            if p.dim() == 4:
                p.mul_(scale)
        g.conv4.bias.fill_(bias)


def sd_to_params(sd, grad=False, d=False):
    """state_dict -> oracle parameter dict (detached fp32 clones; leaves if grad=True)"""
    P = {}
    for k, v in sd.items():
        t = v.detach().clone().float()
        if grad and (k.endswith(".weight") or k.endswith(".bias") or k.endswith("weight_orig")):
            t.requires_grad_(True)
        P[k] = t
    return P


def free_port() -> int:
    """a TCP port nobody listens on right now (bind to 0, read it back): rendezvous ports that do not collide under parallel runs"""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def rounded_weights(P, dtype):
    """The oracle's parameters as a 16-bit kernel sees them: every conv weight (4-D) rounded through ``dtype``, biases left fp32 (the
    kernels add them in fp32).  With these the oracle differs from a 16-bit HIP run only by the 16-bit rounding of the stored
    activations and the accumulation order -- the method of tests/test_kernels_gpu.py, applied to whole networks."""
    return {k: (v.detach().to(dtype).to(torch.float32) if v.dim() == 4 else v.detach().clone()) for k, v in P.items()}


def pinned_vgg(golden_dir):
    """(fixture, torchvision vgg19 state_dict restricted to ``features.*``) when tools/pin_vgg.py's fixture exists AND the environment
    variable SRGANFD_VGG19_WEIGHTS names the weights it was made from (SHA-256 of the features checked); otherwise pytest.skip.
    Neither exists in the build container or on the GPU box (no torchvision, no ImageNet weights offline): the tests that call this
    are how row A7 of SURVEY section 8 becomes pinned, without a code change, wherever both are available."""
    import hashlib
    import pytest
    path = os.path.join(golden_dir, "vgg19_taps.npz")
    wpath = os.environ.get("SRGANFD_VGG19_WEIGHTS", "")
    if not os.path.exists(path):
        pytest.skip("tests/golden/vgg19_taps.npz absent: run tools/pin_vgg.py where torchvision + the ImageNet VGG-19 weights exist")
    if not wpath or not os.path.exists(wpath):
        pytest.skip("SRGANFD_VGG19_WEIGHTS does not name a torchvision vgg19 state_dict")
    g = np.load(path, allow_pickle=False)
    sd = torch.load(wpath, map_location="cpu")
    sd = {k: v for k, v in sd.get("state_dict", sd).items() if k.startswith("features.")}
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].detach().cpu().contiguous().numpy().tobytes())
    if h.hexdigest() != str(g["features_sha256"]):
        pytest.skip("SRGANFD_VGG19_WEIGHTS is not the file the fixture was made from (features SHA-256 differs)")
    return g, sd, wpath
