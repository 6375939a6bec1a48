"""Mirror of the scene-assembly half of utils/visualize.py (prepare_mesh :16-88, prepare_materials :91-108, render_image
:258-285, render_image_with_RT :288-319).  The turntable / gif helpers are offline visualisation (out of scope)."""
import torch
import torch.nn.functional as F

from .. import ops
from ..renderer.pbr_materials import PBRMaterials
from ..structures import Meshes, TexturesUV
from ..topology import subdivide_topology  # noqa: F401  (re-export)


class MeshSubdivider:
    """What `get_mesh_subdivider` returns (optimize_sequence.py:67-89): the static topology of the 4-way subdivided template,
    computed ONCE (SubdivideMeshes precomputes the face table too; PyTorch3D still rebuilds edges per call)."""

    def __init__(self, faces0, n_verts0, device):
        from ..synth import build_topology
        import numpy as np
        t = build_topology(np.asarray(faces0.detach().cpu() if torch.is_tensor(faces0) else faces0), n_verts0)
        self.topo = ops.DeviceTopology(t, torch.zeros(1, 2), torch.zeros(t["faces"].shape[0], 3, dtype=torch.int32), device)
        self.faces = self.topo.faces.long()

    def __call__(self, verts_mm_or_m):
        raise TypeError("call prepare_mesh(..., mesh_subdivider=this) — subdivision is fused into the mesh-prep kernels")


_RAW_TOPO = {}


def raw_topology(mesh_faces, n_verts, device):
    """static tables of the UN-subdivided template (prepare_mesh with mesh_subdivider=None, utils/visualize.py:51-56), built once"""
    import hashlib
    f = torch.as_tensor(mesh_faces).detach().cpu().reshape(-1, 3).contiguous()
    # keyed on the CONTENT of the face table (18 KB for MANO): an address can be reused by another tensor after this one is freed
    key = (hashlib.sha1(f.numpy().tobytes()).hexdigest(), int(n_verts), str(device))
    if key not in _RAW_TOPO:
        from ..synth import build_raw_topology
        if len(_RAW_TOPO) >= 8:                           # a handful of templates at most (hand, arm): bounded
            _RAW_TOPO.pop(next(iter(_RAW_TOPO)))
        t = build_raw_topology(f.numpy(), int(n_verts))
        _RAW_TOPO[key] = ops.DeviceTopology(t, torch.zeros(1, 2), torch.zeros(t["faces"].shape[0], 3, dtype=torch.int32), device)
    return _RAW_TOPO[key]


def prepare_mesh(params, fid, mano_layer, verts_textures, mesh_subdivider, global_pose, configs, device="cuda", vis_normal=False,
                 shared_texture=True, use_arm=False):
    """utils/visualize.py:16-88 for the MANO + UV-texture path HARP runs (verts_textures=False, shared_texture=True, model_type
    'harp').  Returns (hand_joints (B,21,3) m, hand_verts (B,V,3) m, faces (B,F,3), textures)."""
    if configs.get("model_type", "harp") != "harp" or verts_textures:
        raise NotImplementedError("the 'harp' model type with UV textures is the path in scope (SURVEY.md §8)")
    fid = torch.as_tensor(fid).long()
    B = fid.shape[0]
    pose_batch, rot_batch = params["pose"][fid.to(params["pose"].device)], params["rot"][fid.to(params["rot"].device)]   # :26-27 (global_pose forced False, :20)
    trans_batch = params["trans"][fid.to(params["trans"].device)].to(device)
    if use_arm:
        hand_verts, hand_joints = mano_layer(betas=params["shape"].repeat([B, 1]).to(device), global_orient=rot_batch.to(device),
                                             transl=trans_batch, right_hand_pose=pose_batch.to(device),
                                             right_wrist_pose=params["wrist_pose"][fid.to(params["wrist_pose"].device)].to(device),
                                             return_type="mano_w_arm")                                                   # :37-40
    else:
        hand_verts, hand_joints = mano_layer(torch.cat((rot_batch, pose_batch), 1).to(device), params["shape"].repeat([B, 1]).to(device),
                                             trans_batch)                                                                # :42-44
    hand_joints = hand_joints / 1000.0                                                                                   # :46
    if mesh_subdivider is None:                                                                                          # :51-56: the raw template mesh (config C1)
        topo = raw_topology(params["mesh_faces"], hand_verts.shape[1], device)
        vs = hand_verts / 1000.0                                                                                         # :45
    else:
        topo = mesh_subdivider.topo
        vs = ops.subdivide(hand_verts, topo, 1.0 / 1000.0)                                                              # :45, :50-52
    disp = params["verts_disps"]
    if disp is not None:
        if disp.shape[1] != 1:
            raise NotImplementedError("VERT_DISPS_NORMALS=True in the reference (optimize_sequence.py:326): displacement along normals")
        _, hand_verts = ops.normals_displace(vs, disp.to(device), topo)                                                  # :58-64
    else:
        hand_verts = vs
    faces = topo.faces.long()[None].expand(B, -1, -1)
    faces._harp_topo = topo
    uv_map = params["texture"][None, 0].repeat(B, 1, 1, 1).to(device) if shared_texture else params["texture"].to(device)   # :81-83
    textures = TexturesUV(maps=uv_map, faces_uvs=params["faces_uvs"], verts_uvs=params["verts_uvs"])
    return hand_joints, hand_verts, faces, textures


def prepare_materials(params, batch_size, shared_texture=True, device="cuda"):
    """utils/visualize.py:91-108"""
    normal_maps = None
    if "normal_map" in params:
        nm = params["normal_map"][None, 0].repeat(batch_size, 1, 1, 1).to(device) if shared_texture else params["normal_map"].to(device)
        nm = F.normalize(nm, dim=-1)
        normal_maps = TexturesUV(maps=nm, faces_uvs=params["faces_uvs"], verts_uvs=params["verts_uvs"])
    return {"normal_maps": normal_maps}


def _cam_RT(cam, batch_size, img_size, focal_length, device):
    camera_t = torch.stack([-cam[:, 1], -cam[:, 2], 2 * focal_length / (img_size * cam[:, 0] + 1e-9)], dim=1).to(device)   # :268
    R = torch.tensor([[-1., 0., 0.], [0., -1., 0.], [0., 0., 1.]], device=device).repeat(batch_size, 1, 1)                    # :271
    return R, camera_t


def render_image(mesh, cam, batch_size, renderer, img_size, focal_length, silhouette=False, device="cuda", materials_properties=dict()):
    """utils/visualize.py:258-285"""
    materials = PBRMaterials(device=device, shininess=0.0, **materials_properties)
    R, T = _cam_RT(cam, batch_size, img_size, focal_length, device)
    img = renderer(mesh, principal_point=torch.Tensor([(img_size / 2., img_size / 2.)]), focal_length=focal_length, T=T, R=R,
                   materials=materials, image_size=torch.Tensor([(img_size, img_size)]))
    return img[:, :, :, 3] if silhouette else img[:, :, :, 0:3]


def render_image_with_RT(mesh, light_t, light_r, cam_t, cam_r, batch_size, renderer, img_size, focal_length, silhouette=False,
                         materials_properties=dict(), device="cuda"):
    """utils/visualize.py:288-319"""
    materials = PBRMaterials(device=device, shininess=0.0, **materials_properties)
    img = renderer(mesh, principal_point=torch.Tensor([(img_size / 2., img_size / 2.)]), focal_length=focal_length, T=light_t.to(device),
                   R=light_r.to(device), cam_T=cam_t.to(device), cam_R=cam_r.to(device), materials=materials,
                   image_size=torch.Tensor([(img_size, img_size)]))
    return img[:, :, :, 3] if silhouette else img[:, :, :, 0:3]
