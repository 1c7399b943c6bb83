"""Test helper: reproduce the reference initialisation sequence from a seed.

The golden fixtures store checksums instead of the 59 MB of VGG weights; the weights are
reproduced by constructing the same torch.nn containers in the same order as
``/root/reference/daod/modeling/meta_arch/vgg.py:10-24`` (make_layers) and re-initialising
them like ``:102-113`` (_initialize_weights) after ``torch.manual_seed(seed)``.
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

VGG16 = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]
STAGE_SLICES = [(0, 7), (7, 14), (14, 24), (24, 34), (34, 44)]


def reference_vgg_state(seed):
    torch.manual_seed(seed)
    layers = []
    cin = 3
    for v in VGG16:
        if v == "M":
            layers.append(nn.MaxPool2d(2, 2))
        else:
            layers += [nn.Conv2d(cin, v, 3, padding=1), nn.BatchNorm2d(v), nn.ReLU(inplace=True)]
            cin = v
    for m in layers:
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)
    sd = OrderedDict()
    for s, (a, b) in enumerate(STAGE_SLICES):
        for local, m in enumerate(layers[a:b]):
            for k, v in m.state_dict().items():
                sd[f"backbone.vgg{s}.{local}.{k}"] = v.detach().clone()
    return sd


def checksum(t):
    t = t.detach().double().flatten()
    return np.array([t.sum().item(), t.abs().sum().item(),
                     (t * torch.arange(1, t.numel() + 1, dtype=torch.float64) % 7).sum().item()])
