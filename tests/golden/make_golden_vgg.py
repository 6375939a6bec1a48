"""Generates tests/golden/vgg_ref.npz: outputs of the REFERENCE's own perceptual-feature module (model/vgg.py:10-56, imported from
/root/reference) on seeded inputs.

    python tests/golden/make_golden_vgg.py

`model/vgg.py` does `from torchvision import models; models.vgg16(pretrained=True).features` — torchvision is not installed in the build
image and the pretrained file cannot be downloaded.  The ONE thing stubbed is that constructor: a module object `torchvision.models` whose
`vgg16(pretrained=...)` returns an object with `.features` = the published VGG16 layer sequence (tests/golden/vgg_filters.py) holding SEEDED
filters.  Everything the fixture pins is the reference's own code: which feature indices go into which slice (0:4, 4:9, 9:16, 16:23), the
four taps, `layers_weights`, the flatten / concatenation order of its forward, requires_grad=False.  The fixture is data (inputs and the
module's output rows); no reference source is stored."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from vgg_filters import vgg16_features  # noqa: E402

tv, models = types.ModuleType("torchvision"), types.ModuleType("torchvision.models")
models.vgg16 = lambda pretrained=False, **kw: types.SimpleNamespace(features=vgg16_features())
tv.models = models
sys.modules["torchvision"], sys.modules["torchvision.models"] = tv, models
sys.path.insert(0, "/root/reference")
from model.vgg import Vgg16Features  # noqa: E402  (the reference's module)

S, N = 16, 2
LW = [1, 1 / 16, 1 / 8, 1 / 4, 1]                                   # optimize_sequence.py:405
g = torch.Generator().manual_seed(77)
x = torch.rand(N, 3, S, S, generator=g)
with torch.no_grad():
    ref_default = Vgg16Features()                                   # layers_weights=None -> [1/32, 1/16, 1/8, 1/4, 1]
    ref_fit = Vgg16Features(layers_weights=LW)
    y_default, y_fit = ref_default(x), ref_fit(x)
assert all(not p.requires_grad for p in ref_fit.parameters())
keys = sorted(ref_fit.state_dict().keys())
np.savez_compressed(os.path.join(HERE, "vgg_ref.npz"), x=x.numpy(), y_default=y_default.numpy(), y_fit=y_fit.numpy(),
                    layers_weights_default=np.asarray(ref_default.layers_weights, np.float64), layers_weights_fit=np.asarray(LW, np.float64),
                    state_dict_keys=np.asarray(keys))
print("wrote vgg_ref.npz:", y_fit.shape, keys[:4], "...")
