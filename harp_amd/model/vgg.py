"""Perceptual features for the appearance stage: mirror of the reference's `model/vgg.py:10-56`.

`Vgg16Features(x)` for x (N,3,H,W) returns ONE row per image: the concatenation of the flattened input and of the flattened
activations relu1_2 / relu2_2 / relu3_3 / relu4_3 of VGG16, each scaled by its `layers_weights` entry; the loop's term is
`L1(vgg(y_pred * mask), vgg(y_true * mask))` (optimize_sequence.py:546-547, weight 1.0, inputs in [0,1] without ImageNet
normalisation).

The reference takes the layers from `torchvision.models.vgg16(pretrained=True).features[0:23]`; torchvision is not a dependency of
this package, so the 10 convolutions are declared here (same layer indices -> same state-dict keys as the reference module:
`slice1.0.weight`, `slice1.2.weight`, `slice2.5.weight`, ...) and the weights come from a file:

    Vgg16Features(layers_weights=[1, 1/16, 1/8, 1/4, 1], weights="vgg16-397923af.pth")   # torchvision's vgg16 state dict
    Vgg16Features(weights=reference_module.state_dict())                                  # or the reference module's own
    Vgg16Features(weights="random")                                                       # tests only: seeded random filters

With `weights=None` the module tries torchvision (as the reference does) and raises if it is absent: there is no silent
random-weight fallback.  What is pinned: the module structure, row layout and weighting against rows written by the imported reference
module (tests/golden/vgg_ref.npz, tests/golden/make_golden_vgg.py); what is not: torchvision's PRETRAINED filters (absent here).

On a HIP tensor `forward` runs the ten convolutions, the pools and — under autograd — their data gradients on this repo's matrix-core
kernels (csrc/conv.hip through `model/vgg_hip.py:Vgg16Rows`), so the reference loop body's `l1_loss(vgg(a), vgg(b))`
(optimize_sequence.py:546-547) lands on them; `FitEngine.set_perceptual` runs the same kernels as one fused 21-launch term
(harp_vgg16_term).  A CPU tensor goes through the torch layers declared here: that path exists for the CPU tests of the layout.
"""
import torch

# torchvision vgg16.features[0:23]: index -> (in, out) of the 3x3 convolutions; 4 / 9 / 16 are the 2x2 max-pools
_CONVS = {0: (3, 64), 2: (64, 64), 5: (64, 128), 7: (128, 128), 10: (128, 256), 12: (256, 256), 14: (256, 256), 17: (256, 512),
          19: (512, 512), 21: (512, 512)}
_POOLS = (4, 9, 16)
_SLICES = ((0, 4), (4, 9), (9, 16), (16, 23))


def _layer(ix):
    if ix in _CONVS:
        return torch.nn.Conv2d(*_CONVS[ix], kernel_size=3, padding=1)
    if ix in _POOLS:
        return torch.nn.MaxPool2d(kernel_size=2, stride=2)
    return torch.nn.ReLU(inplace=False)


def feature_length(H, W):
    """row length of `Vgg16Features.forward` for an (N,3,H,W) input (H, W multiples of 8)"""
    return H * W * 3 + H * W * 64 + (H // 2) * (W // 2) * 128 + (H // 4) * (W // 4) * 256 + (H // 8) * (W // 8) * 512


class Vgg16Features(torch.nn.Module):
    def __init__(self, requires_grad=False, layers_weights=None, weights=None, seed=0):
        super().__init__()
        self.layers_weights = [1 / 32, 1 / 16, 1 / 8, 1 / 4, 1] if layers_weights is None else list(layers_weights)
        for n, (lo, hi) in enumerate(_SLICES, start=1):
            seq = torch.nn.Sequential()
            for ix in range(lo, hi):
                seq.add_module(str(ix), _layer(ix))
            setattr(self, f"slice{n}", seq)
        self._load(weights, seed)
        if not requires_grad:
            for p in self.parameters():
                p.requires_grad = False

    def _load(self, weights, seed):
        if isinstance(weights, str) and weights == "random":
            g = torch.Generator().manual_seed(seed)
            with torch.no_grad():
                for p in self.parameters():
                    if p.dim() == 4:        # He-scaled filters keep the activations O(1) through the 10 layers
                        p.copy_(torch.randn(p.shape, generator=g) * (2.0 / (p.shape[1] * 9)) ** 0.5)
                    else:
                        p.zero_()
            return
        if weights is None:
            try:
                from torchvision import models
            except ImportError as e:
                raise RuntimeError("Vgg16Features needs the pretrained VGG16 filters: pass weights=<path to torchvision's vgg16 state dict> "
                                   "(torchvision is not installed, nothing can be downloaded)") from e
            weights = models.vgg16(pretrained=True).state_dict()
        if isinstance(weights, (str, bytes)) or hasattr(weights, "__fspath__"):
            weights = torch.load(weights, map_location="cpu")
        own = self.state_dict()
        slice_of = {ix: n for n, (lo, hi) in enumerate(_SLICES, start=1) for ix in range(lo, hi)}
        picked = {}
        for k, v in weights.items():
            parts = k.split(".")
            if parts[0] == "features" and parts[1].isdigit() and int(parts[1]) in _CONVS:      # torchvision layout
                picked[f"slice{slice_of[int(parts[1])]}.{parts[1]}.{parts[2]}"] = v
            elif k in own:                                                                     # the reference module's layout
                picked[k] = v
        missing = sorted(set(own) - set(picked))
        if missing:
            raise KeyError(f"VGG16 state dict lacks {missing}")
        self.load_state_dict(picked)

    def features(self, x, skip_input=False, weighted=True):
        """the five flattened maps as a list (what `forward` concatenates); skip_input drops the first (the image itself);
        weighted=False leaves the `layers_weights` factors to the caller (no scaled copies of the activations)"""
        w = self.layers_weights if weighted else [None] * 5
        scale = (lambda t, f: t if f is None else f * t)
        feats = [] if skip_input else [scale(x.flatten(start_dim=1), w[0])]
        h = x
        for n in range(1, 5):
            h = getattr(self, f"slice{n}")(h)
            feats.append(scale(h.flatten(start_dim=1), w[n]))
        return feats

    hip_precision = 0                # 0: float32 MFMA (parity anchor), 1: three-term bf16 split (model/conv_hip.py)

    def _hip(self, device):
        from .vgg_hip import Vgg16Hip
        key = (str(device), int(self.hip_precision))
        cache = self.__dict__.setdefault("_hip_cache", {})
        if key not in cache:
            cache[key] = Vgg16Hip(self, device, self.hip_precision)
        return cache[key]

    def forward(self, x):
        if x.is_cuda:
            from .vgg_hip import Vgg16Rows
            return Vgg16Rows.apply(x, self._hip(x.device))
        return torch.cat(self.features(x), 1)
