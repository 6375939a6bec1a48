"""arap_loss (loss/arap.py:4-57) and the two PyTorch3D mesh regularisers of the loop (mesh_laplacian_smoothing,
mesh_normal_consistency: optimize_sequence.py:536-537) on ONE gather-style HIP kernel (csrc/losses.hip: mesh_reg_kernel)."""
import torch

from .. import _lib


class _MeshReg(torch.autograd.Function):
    @staticmethod
    def forward(ctx, verts, ref_verts, topo):
        verts = verts.contiguous().float()
        ref = ref_verts.contiguous().float() if ref_verts is not None else None
        B, V, _ = verts.shape
        loss = torch.zeros(3, dtype=torch.float32, device=verts.device)
        _lib.check(_lib.lib().harp_mesh_regularizers(_lib.ptr(verts), _lib.ptr(ref), _lib.ptr(topo.nbr_off), _lib.ptr(topo.nbr_idx),
                                                     _lib.ptr(topo.nc_pairs), _lib.ptr(topo.vp_off), _lib.ptr(topo.vp_idx), B, V,
                                                     topo.nc_pairs.shape[0], topo.E, None, _lib.ptr(loss), None, _lib.stream()),
                   "harp_mesh_regularizers")
        ctx.save_for_backward(verts, ref)
        ctx.topo = topo
        return loss

    @staticmethod
    def backward(ctx, g):
        verts, ref = ctx.saved_tensors
        topo = ctx.topo
        B, V, _ = verts.shape
        gv = torch.zeros_like(verts)
        scratch = torch.zeros(3, dtype=torch.float32, device=verts.device)
        _lib.check(_lib.lib().harp_mesh_regularizers(_lib.ptr(verts), _lib.ptr(ref), _lib.ptr(topo.nbr_off), _lib.ptr(topo.nbr_idx),
                                                     _lib.ptr(topo.nc_pairs), _lib.ptr(topo.vp_off), _lib.ptr(topo.vp_idx), B, V,
                                                     topo.nc_pairs.shape[0], topo.E, _lib.ptr(g.contiguous().float()), _lib.ptr(scratch),
                                                     _lib.ptr(gv), _lib.stream()), "harp_mesh_regularizers")
        return gv, None, None


def mesh_regularizers(meshes, ref_meshes=None):
    """-> tensor [laplacian, normal_consistency, arap] (arap = 0 without ref_meshes)"""
    ref = ref_meshes.verts_padded()[0] if ref_meshes is not None else None
    return _MeshReg.apply(meshes.verts_padded(), ref, meshes.topo)


def mesh_laplacian_smoothing(meshes, method="uniform"):
    if method != "uniform":
        raise NotImplementedError("HARP uses the default uniform Laplacian (optimize_sequence.py:536)")
    return mesh_regularizers(meshes)[0]


def mesh_normal_consistency(meshes):
    return mesh_regularizers(meshes)[1]


def arap_loss(meshes, ref_meshes, target_length: float = 0.0):
    """loss/arap.py:4-57; the reference mesh is the first (only) mesh of `ref_meshes`, extended to the batch (:36-37)."""
    if meshes.isempty():
        return torch.tensor([0.0], dtype=torch.float32, device=meshes.device, requires_grad=True)      # arap.py:25-28
    return mesh_regularizers(meshes, ref_meshes)[2]
