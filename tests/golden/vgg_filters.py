"""Seeded stand-in filters in torchvision's vgg16 `features` layout (configuration "D" of Simonyan & Zisserman: 13 3x3 convolutions, 5
max pools; `torchvision.models.vgg16().features` indices) — shared by the fixture generator (tests/golden/make_golden_vgg.py) and the tests
that read the fixture.  The pretrained filters cannot exist in the build image (no network): these are He-scaled draws, one generator per
layer, so the same numbers come out wherever this file runs with the same torch version."""
import torch

CFG_D = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]


def vgg16_features(seed=1234):
    """torch.nn.Sequential with torchvision's layer order: Conv2d(3x3, pad 1), ReLU(inplace=True), ..., MaxPool2d(2, 2)"""
    layers, cin = [], 3
    for v in CFG_D:
        if v == "M":
            layers.append(torch.nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            conv = torch.nn.Conv2d(cin, v, kernel_size=3, padding=1)
            g = torch.Generator().manual_seed(seed + len(layers))
            with torch.no_grad():
                conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (2.0 / (9 * cin)) ** 0.5)
                conv.bias.copy_(torch.randn(v, generator=g) * 0.05)
            layers += [conv, torch.nn.ReLU(inplace=True)]
            cin = v
    return torch.nn.Sequential(*layers)


def state_dict_torchvision_layout(seed=1234):
    return {f"features.{k}": v for k, v in vgg16_features(seed).state_dict().items()}
