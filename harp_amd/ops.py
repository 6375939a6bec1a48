"""torch.autograd bindings of the HIP kernels (C-ABI in include/harp_hip.h).

These are the building blocks the reference-API mirror modules (renderer/, utils/, loss/, manopth/) and the
fused engine call.  Every op takes/returns plain contiguous float32/int32 HIP tensors."""
import math

import torch

from . import _lib

SIL_SIGMA = 1e-7                                        # optimize_sequence.py:426
SIL_BLUR = math.log(1.0 / 1e-4 - 1.0) * SIL_SIGMA       # renderer_helper.py:46


def rasterize_workspace(B, F, S, device):
    n = _lib.lib().harp_rasterize_ws_bytes(B, F, S)
    return torch.empty(n, dtype=torch.uint8, device=device)


def rasterize_fwd(ndc, faces, S, soft=False, blur_radius=0.0, sigma=1.0, want_zbuf=True, ws=None):
    """ndc (B,V,3) f32, faces (F,3) i32 -> face_id (B,S,S) i32, zbuf (B,S,S)|None, alpha (B,S,S)|None, ws."""
    B, V, _ = ndc.shape
    F = faces.shape[0]
    dev = ndc.device
    if ws is None:
        ws = rasterize_workspace(B, F, S, dev)
    face_id = torch.empty(B, S, S, dtype=torch.int32, device=dev)
    zbuf = torch.empty(B, S, S, dtype=torch.float32, device=dev) if want_zbuf else None
    alpha = torch.empty(B, S, S, dtype=torch.float32, device=dev) if soft else None
    rc = _lib.lib().harp_rasterize_fwd(_lib.ptr(ndc), _lib.ptr(faces), B, V, F, S, int(soft), blur_radius, sigma,
                                       _lib.ptr(ws), _lib.ptr(face_id), _lib.ptr(zbuf), _lib.ptr(alpha), _lib.stream())
    _lib.check(rc, "harp_rasterize_fwd")
    return face_id, zbuf, alpha, ws


def silhouette_bwd(faces, V, S, blur_radius, sigma, ws, alpha, g_alpha, g_ndc):
    B = alpha.shape[0]
    rc = _lib.lib().harp_silhouette_bwd(_lib.ptr(faces), B, V, faces.shape[0], S, blur_radius, sigma, _lib.ptr(ws),
                                        _lib.ptr(alpha), _lib.ptr(g_alpha.contiguous()), _lib.ptr(g_ndc), _lib.stream())
    _lib.check(rc, "harp_silhouette_bwd")


class _SoftSilhouette(torch.autograd.Function):
    """alpha = SoftSilhouetteShader(MeshRasterizer(K=50, blur))(mesh)[..., 3] (renderer_helper.py:44-58)."""

    @staticmethod
    def forward(ctx, ndc, faces, S, blur_radius, sigma):
        ndc = ndc.contiguous()
        face_id, _, alpha, ws = rasterize_fwd(ndc, faces, S, soft=True, blur_radius=blur_radius, sigma=sigma, want_zbuf=False)
        ctx.save_for_backward(faces, alpha, ws)
        ctx.meta = (ndc.shape, S, blur_radius, sigma)
        ctx.mark_non_differentiable(face_id)
        return alpha, face_id

    @staticmethod
    def backward(ctx, g_alpha, _g_face):
        faces, alpha, ws = ctx.saved_tensors
        shape, S, blur_radius, sigma = ctx.meta
        g_ndc = torch.zeros(shape, dtype=torch.float32, device=alpha.device)
        silhouette_bwd(faces, shape[1], S, blur_radius, sigma, ws, alpha, g_alpha, g_ndc)
        return g_ndc, None, None, None, None


def soft_silhouette(ndc, faces, S, blur_radius=SIL_BLUR, sigma=SIL_SIGMA):
    return _SoftSilhouette.apply(ndc, faces, S, blur_radius, sigma)
