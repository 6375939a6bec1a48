"""Bindings of the matrix-core 3x3 convolution (csrc/conv.hip, `harp_conv3x3*` in include/harp_hip.h): the building block of the
perceptual term's VGG16 stack (reference: model/vgg.py:10-56).  Activations are NHWC float32 HIP tensors; filters are packed once
per precision (0 = float32 MFMA, 1 = three-term bf16 split)."""
import ctypes

import torch

from .. import _lib

RELU, RELU_TAP, GATE, UNPOOL = 0, 1, 2, 3
F32, BF16X3 = 0, 1


def _pad(c, m):
    return (c + m - 1) // m * m


def pack_filters(w, precision=F32, transpose=False):
    """w (Cout,Cin,3,3) float32 HIP tensor in torch's layout -> packed filter slabs (uint8 tensor).  transpose: the filters of the
    data gradient (a convolution over the output gradient, Cout input channels -> Cin output channels)."""
    Cout, Cin = w.shape[:2]
    out_c, in_c = (Cin, Cout) if transpose else (Cout, Cin)
    L = _lib.lib()
    packed = torch.zeros(L.harp_conv3x3_filter_bytes(out_c, in_c), dtype=torch.uint8, device=w.device)
    w = w.contiguous().float()
    _lib.check(L.harp_conv3x3_pack_filters(_lib.ptr(w), Cout, Cin, int(transpose), precision, _lib.ptr(packed), _lib.stream()), "harp_conv3x3_pack_filters")
    return packed


def conv3x3(x, filters, Cout, bias=None, epilogue=RELU, precision=F32, out=None, pooled=None, target=None, target_row=None, tap_scale=0.0,
            g_tap=None, loss=None, gate=None):
    """one launch of harp_conv3x3 on x (N,H,W,Cin); the optional tensors are the epilogue's operands (include/harp_hip.h)"""
    N, H, W, Cin = x.shape
    a = _lib.Conv3x3Args()
    a.in_, a.filters, a.bias, a.out, a.pooled = _lib.ptr(x), _lib.ptr(filters), _lib.ptr(bias), _lib.ptr(out), _lib.ptr(pooled)
    a.target, a.target_row, a.g_tap, a.loss, a.gate = _lib.ptr(target), _lib.ptr(target_row), _lib.ptr(g_tap), _lib.ptr(loss), _lib.ptr(gate)
    a.N, a.H, a.W, a.Cin, a.Cout, a.precision, a.epilogue, a.tap_scale = N, H, W, Cin, Cout, precision, epilogue, float(tap_scale)
    _lib.check(_lib.lib().harp_conv3x3(ctypes.byref(a), _lib.stream()), "harp_conv3x3")
