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
