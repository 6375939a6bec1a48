"""The perceptual term on the HIP convolution kernels (csrc/conv.hip): `L1Loss(vgg(y_pred * mask), vgg(y_true * mask))` of
optimize_sequence.py:546-547 with `vgg` = model/vgg.py's Vgg16Features, forward and data gradient, as `harp_vgg16_term` runs it.

`Vgg16Hip(module)` takes the filters of a `harp_amd.model.vgg.Vgg16Features` (same state-dict keys as the reference module) and packs
them for the matrix cores; nothing here runs a convolution through torch."""
import ctypes

import torch

from .. import _lib
from . import conv_hip
from .vgg import _CONVS, _SLICES, feature_length

_ORDER = sorted(_CONVS)                      # 0, 2, 5, 7, 10, 12, 14, 17, 19, 21: vgg16.features indices of the ten convolutions


def tap_shapes(S):
    """(H, W, C) of the four tap activations relu1_2 ... relu4_3 for an S x S image, NHWC"""
    return [(S, S, 64), (S // 2, S // 2, 128), (S // 4, S // 4, 256), (S // 8, S // 8, 512)]


class Vgg16Hip:
    def __init__(self, module, device, precision=conv_hip.F32):
        self.dev = torch.device(device)
        self.precision = int(precision)
        self.layers_weights = [float(w) for w in module.layers_weights]
        sd = module.state_dict()
        slice_of = {ix: n for n, (lo, hi) in enumerate(_SLICES, start=1) for ix in range(lo, hi)}
        self._keep = []                          # device tensors the C struct points into
        net = _lib.Vgg16()
        for k, ix in enumerate(_ORDER):
            w = sd[f"slice{slice_of[ix]}.{ix}.weight"].detach().to(self.dev).float().contiguous()
            b = sd[f"slice{slice_of[ix]}.{ix}.bias"].detach().to(self.dev).float().contiguous()
            f = conv_hip.pack_filters(w, self.precision)
            ft = conv_hip.pack_filters(w, self.precision, transpose=True) if k > 0 else None
            self._keep += [w, b, f, ft]
            net.filters[k], net.filters_t[k], net.bias[k] = _lib.ptr(f), _lib.ptr(ft), _lib.ptr(b)
            if k == 0:      # (64,3,3,3) -> (9,3,64) with the taps mirrored: the layout the last backward kernel reads with scalar loads
                w0t = w.reshape(64, 3, 9).flip(2).permute(2, 1, 0).contiguous()
                self._keep.append(w0t)
                net.w0t = _lib.ptr(w0t)
        for i, w in enumerate(self.layers_weights):
            net.layer_w[i] = w
        net.precision = self.precision
        self.net = net
        self._ws = {}

    def workspace(self, N, S, with_gradient):
        key = (N, S, bool(with_gradient))
        if key not in self._ws:
            n = _lib.lib().harp_vgg16_ws_bytes(N, S, int(with_gradient))
            if n == 0:
                raise ValueError(f"perceptual term: image size {S} is not a multiple of 8")
            self._ws[key] = torch.zeros(n, dtype=torch.uint8, device=self.dev)
        return self._ws[key]

    def features(self, image, mask, rows=None, N=None, out=None):
        """tap activations of image[rows] * mask[rows] (image (T,S,S,3), mask (T,S,S); rows int32 (N,) or None = the first N): list of
        four NHWC tensors (written into `out` when given)"""
        S = image.shape[1]
        N = int(rows.shape[0]) if rows is not None else (image.shape[0] if N is None else N)
        if out is None:
            out = [torch.empty((N,) + s, device=self.dev) for s in tap_shapes(S)]
        ws = self.workspace(N, S, False)
        rc = _lib.lib().harp_vgg16_features(ctypes.byref(self.net), _lib.ptr(image), _lib.ptr(mask), _lib.ptr(rows), N, S, _lib.ptr(ws),
                                            *[_lib.ptr(o) for o in out], _lib.stream())
        _lib.check(rc, "harp_vgg16_features")
        return out

    def term(self, rgb, y_true, mask, rows, target, target_by_row, g_rgb, loss, weight=1.0, covered=None):
        """enqueue the whole term: *loss (a float32 HIP scalar / 1-element view) = the term, g_rgb updated in place (include/harp_hip.h)"""
        N, S = rgb.shape[0], rgb.shape[1]
        t = _lib.Vgg16TermArgs()
        t.rgb, t.y_true, t.mask, t.rows = _lib.ptr(rgb), _lib.ptr(y_true), _lib.ptr(mask), _lib.ptr(rows)
        for k in range(4):
            t.target[k] = _lib.ptr(target[k])
        t.target_by_row, t.covered, t.g_rgb, t.weight, t.loss = int(target_by_row), _lib.ptr(covered), _lib.ptr(g_rgb), float(weight), _lib.ptr(loss)
        t.N, t.S, t.ws = N, S, _lib.ptr(self.workspace(N, S, True))
        _lib.check(_lib.lib().harp_vgg16_term(ctypes.byref(self.net), ctypes.byref(t), _lib.stream()), "harp_vgg16_term")


__all__ = ["Vgg16Hip", "tap_shapes", "feature_length"]
