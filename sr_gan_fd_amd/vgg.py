"""VGG-19 ``features[0:36]`` parameter container (torchvision layout, restated from the public
architecture: torchvision is a third-party dependency of the reference, absent from its tree).
State-dict keys match torchvision's ``vgg19().features`` (``{idx}.weight`` / ``{idx}.bias``)."""
from __future__ import annotations

import torch
from torch import nn

CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512)


def build_vgg19_features(weights_path: str = "", seed: int = 19) -> nn.Sequential:
    layers, cin = [], 3
    for v in CFG:
        if v == "M":
            layers.append(nn.MaxPool2d(2, 2))
        else:
            layers += [nn.Conv2d(cin, v, 3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    feats = nn.Sequential(*layers)
    if weights_path:
        sd = torch.load(weights_path, map_location="cpu")
        sd = sd.get("state_dict", sd)
        own = {k[len("features."):]: v for k, v in sd.items() if k.startswith("features.")} or sd
        feats.load_state_dict({k: v for k, v in own.items() if k in feats.state_dict()}, strict=True)
    else:
        # no ImageNet weights offline: deterministic He init so that content-loss VALUES are reproducible
        # (they are logged only: the reference detaches them, model.py:552)
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for m in feats:
                if isinstance(m, nn.Conv2d):
                    fan_in = m.weight.shape[1] * 9
                    m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5)
                    m.bias.zero_()
    return feats
