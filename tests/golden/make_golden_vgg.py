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
# a second, MULTI-TILE input: 64 x 48 (non-square; 4 x 3 tiles of 16 x 16 at full resolution, one ragged 8 x 6 tile at the last level).  The
# rows are 377 856 floats each: the fixture keeps every 5th element of each row, the absolute sum of each of the five segments and the
# L1 distance of the two rows (the loop's term, optimize_sequence.py:546-547)
H2, W2, STRIDE = 64, 48, 5
x2 = torch.rand(N, 3, H2, W2, generator=g)
with torch.no_grad():
    y2 = ref_fit(x2).double()
seg = np.cumsum([0, 3 * H2 * W2, 64 * H2 * W2, 128 * (H2 // 2) * (W2 // 2), 256 * (H2 // 4) * (W2 // 4), 512 * (H2 // 8) * (W2 // 8)])
assert seg[-1] == y2.shape[1]
# the loop's term and its gradient on a MASKED pair (optimize_sequence.py:546-547): 64 x 64, the mask an off-centre blob that leaves most
# 16 x 16 tiles outside its receptive field (the bounded mode and the shifted tile grids of csrc/conv.hip are checked against these)
S3 = 64
yy, xx = torch.meshgrid(torch.arange(S3), torch.arange(S3), indexing="ij")
mask3 = (((yy - 37.0) / 13.0) ** 2 + ((xx - 21.0) / 9.0) ** 2 < 1.0).float()
a3 = torch.rand(1, S3, S3, 3, generator=g).requires_grad_(True)
b3 = torch.rand(1, S3, S3, 3, generator=g)
loss3 = torch.nn.L1Loss()(ref_fit((a3 * mask3[None, ..., None]).permute(0, 3, 1, 2)), ref_fit((b3 * mask3[None, ..., None]).permute(0, 3, 1, 2)))
(grad3,) = torch.autograd.grad(loss3, a3)
np.savez_compressed(os.path.join(HERE, "vgg_ref.npz"), x=x.numpy(), y_default=y_default.numpy(), y_fit=y_fit.numpy(),
                    pair_pred=a3.detach().numpy(), pair_true=b3.numpy(), pair_mask=mask3.numpy(), pair_loss=np.asarray(loss3.item()),
                    pair_grad=grad3.numpy(),
                    layers_weights_default=np.asarray(ref_default.layers_weights, np.float64), layers_weights_fit=np.asarray(LW, np.float64),
                    state_dict_keys=np.asarray(keys),
                    x_64x48=x2.numpy(), y_64x48_every5th=y2[:, ::STRIDE].float().numpy(), y_64x48_segments=seg,
                    y_64x48_segment_abs_sums=np.asarray([[y2[n, seg[i]:seg[i + 1]].abs().sum().item() for i in range(5)] for n in range(N)]),
                    y_64x48_l1=np.asarray((y2[0] - y2[1]).abs().mean().item()))
print("wrote vgg_ref.npz:", y_fit.shape, y2.shape, keys[:4], "...")
