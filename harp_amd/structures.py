"""Minimal stand-ins for the PyTorch3D containers the reference passes around (`Meshes`, `TexturesUV`:
utils/visualize.py:50-52, 84-87, 99-101; optimize_sequence.py:434, 463).  They only hold tensors + the static device
topology; nothing is recomputed per instance (PyTorch3D rebuilds edge tables / packed views for every `Meshes(...)`)."""
import torch


class TexturesUV:
    def __init__(self, maps, faces_uvs, verts_uvs):
        self.maps, self.faces_uvs, self.verts_uvs = maps, faces_uvs, verts_uvs

    def to(self, device):
        return self

    def maps_padded(self):
        return self.maps


class Meshes:
    """verts (B,V,3) float32 HIP tensor; faces (B,F,3) or (F,3) carrying a `_harp_topo` attribute (set by
    harp_amd.utils.visualize.prepare_mesh / MeshSubdivider) that names the static DeviceTopology."""

    def __init__(self, verts, faces, textures=None, topo=None):
        self._verts = verts
        self._faces = faces
        self.textures = textures
        self.topo = topo if topo is not None else getattr(faces, "_harp_topo", None)
        if self.topo is None:
            raise ValueError("harp_amd.Meshes needs faces produced by prepare_mesh / MeshSubdivider (static topology tables)")
        self.device = verts.device

    def __len__(self):
        return self._verts.shape[0]

    def isempty(self):
        return self._verts.numel() == 0

    def verts_padded(self):
        return self._verts

    def faces_padded(self):
        return self._faces

    def verts_normals_padded(self):
        from . import ops
        return ops.vertex_normals(self._verts, self.topo)

    def extend(self, n):
        return Meshes(self._verts.repeat(n, 1, 1), self._faces, self.textures, self.topo)
