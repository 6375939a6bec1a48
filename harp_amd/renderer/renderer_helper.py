"""Mirror of renderer/renderer_helper.py on the HIP kernels: same factory functions, same renderer call signature
`renderer(mesh, principal_point=, focal_length=, T=, R=, [cam_T=, cam_R=,] materials=, image_size=) -> (B,S,S,4)`
(utils/visualize.py:272-279, 304-313).  No (B,S,S,K) fragments exist: the soft-silhouette alpha and the K=1 hit come out of
one fused raster walk, shading is one fused kernel (see csrc/raster.hip, csrc/shade.hip)."""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .. import ops


def _scalar(x):
    return float(torch.as_tensor(x).reshape(-1)[0])


def _pp(principal_point):
    p = torch.as_tensor(principal_point, dtype=torch.float32).reshape(-1)
    return float(p[0]), float(p[1])


def _size(image_size):
    return int(torch.as_tensor(image_size).reshape(-1)[0])


def look_at_rotation(camera_position, at, up):
    """PyTorch3D look_at_rotation, differentiable torch ops on (B,3) tensors."""
    z = F.normalize(at - camera_position, eps=1e-5)
    x = F.normalize(torch.cross(up.expand_as(z), z, dim=1), eps=1e-5)
    y = F.normalize(torch.cross(z, x, dim=1), eps=1e-5)
    close = torch.isclose(x, torch.zeros((), dtype=x.dtype, device=x.device), atol=5e-3).all(dim=1, keepdim=True)
    if close.any():
        x = torch.where(close, F.normalize(torch.cross(y, z, dim=1), eps=1e-5), x)
    return torch.cat((x[:, None, :], y[:, None, :], z[:, None, :]), dim=1).transpose(1, 2)


class SilhouetteRenderer:
    """MeshRenderer(MeshRasterizer(K=50, blur), SoftSilhouetteShader) (renderer_helper.py:44-58)."""

    def __init__(self, image_size, sigma, faces_per_pixel):
        self.image_size, self.sigma = image_size, sigma
        self.blur_radius = math.log(1.0 / 1e-4 - 1.0) * sigma                                       # renderer_helper.py:46
        # K is not a buffer size in the fused kernel: every face within the blur radius contributes.  The reference keeps the K nearest;
        # ASSUMPTION (checked, not taken on faith: tests/test_gpu_fragments.py::test_depth_complexity_stays_below_the_silhouette_cap
        # counts the fragments per pixel of the bench scenes — hand and arm — with the fragment-level op): fewer than 50 faces ever lie
        # within the blur radius of one pixel of a hand / arm mesh, so a cap of 50 or more never binds and the fused kernel returns what
        # the capped reference returns.  A cap that can bind (K < 50) is honoured exactly through the fragment-level path below, which
        # holds up to 64 slots per pixel; K > 64 cannot bind either while the count stays below 50 and takes the fused kernel too.
        self.faces_per_pixel = faces_per_pixel

    def __call__(self, mesh, principal_point=None, focal_length=None, T=None, R=None, materials=None, image_size=None, **kw):
        S = _size(image_size) if image_size is not None else self.image_size
        dev = mesh.device
        ndc = ops.project(mesh.verts_padded(), R.to(dev), T.to(dev), _scalar(focal_length), S, _pp(principal_point))
        if self.faces_per_pixel < 50:                 # honour a binding cap: K nearest fragments -> sigmoid_alpha_blend (Appendix A.3)
            p2f, _, _, dists = ops.rasterize_fragments(ndc, mesh.topo.faces, S, self.blur_radius, self.faces_per_pixel)
            prob = torch.sigmoid(-dists / self.sigma) * (p2f >= 0).to(dists.dtype)
            alpha = 1.0 - torch.prod(1.0 - prob, dim=-1)
        else:
            alpha, _ = ops.soft_silhouette(ndc, mesh.topo.faces, S, self.blur_radius, self.sigma)
        ones = torch.ones_like(alpha)
        return torch.stack([ones, ones, ones, alpha], -1)                                          # sigmoid_alpha_blend: RGB = 1, A = alpha


class _PhongBase:
    bg = (1.0, 1.0, 1.0)

    def _shade(self, mesh, ndc, face_id, ws, materials, light_pos, colors, S, focal, pp, zl=None, light_R=None, light_T=None):
        tex = mesh.textures
        if tex is None:
            raise ValueError("mesh.textures (TexturesUV) is required")
        topo = mesh.topo
        topo.set_uvs(tex.verts_uvs, tex.faces_uvs)
        vn = mesh.verts_normals_padded()                                                            # renderer_helper.py:495
        nm = materials.normal_maps.maps_padded()[0] if (materials is not None and materials.use_normal_map) else None
        rgb = ops.shade(ndc, mesh.verts_padded(), vn, tex.maps_padded()[0], nm, light_pos, colors, face_id, ws, topo, S, focal,
                        zl=zl, light_R=light_R, light_T=light_T, pp=pp, bg=self.bg)
        a = (face_id >= 0).to(rgb.dtype)            # softmax_rgb_blend alpha for K=1 is sigmoid(-d/sigma) in (0.5,1]; HARP never reads it
        return torch.cat([rgb, a[..., None]], -1)


class PhongRenderer(_PhongBase):
    """MeshRenderer(MeshRasterizer(K=1), SoftPhongShaderPBR) (renderer_helper.py:60-81, 106-190)."""

    def __init__(self, image_size, light_posi):
        self.image_size, self.light_posi = image_size, light_posi

    def __call__(self, mesh, principal_point=None, focal_length=None, T=None, R=None, materials=None, image_size=None, **kw):
        S, focal, pp, dev = _size(image_size), _scalar(focal_length), _pp(principal_point), mesh.device
        ndc = ops.project(mesh.verts_padded(), R.to(dev), T.to(dev), focal, S, pp)
        _, face_id, ws = ops.depth_raster(ndc, mesh.topo.faces, S)
        lp = torch.as_tensor(self.light_posi, dtype=torch.float32, device=dev).reshape(-1, 3).expand(len(mesh), 3)
        colors = torch.tensor([0.5] * 3 + [0.4] * 3 + [0.1] * 3, device=dev)                      # renderer_helper.py:70-73; shininess 0 -> constant specular
        return self._shade(mesh, ndc, face_id, ws, materials, lp, colors, S, focal, pp)


class MeshRendererShadow(_PhongBase):
    """MeshRendererShadow(MeshRasterizer(K=1), SoftPhongShaderShadow) (renderer_helper.py:306-412, 416-451, 526-592)."""

    def __init__(self, image_size, light_posi, amb_ratio):
        self.image_size, self.light_posi, self.amb_ratio = image_size, light_posi, amb_ratio

    def __call__(self, meshes_world, principal_point=None, focal_length=None, T=None, R=None, cam_T=None, cam_R=None, materials=None,
                 image_size=None, **kw):
        mesh = meshes_world
        S, focal, pp, dev = _size(image_size), _scalar(focal_length), _pp(principal_point), mesh.device
        verts = mesh.verts_padded()
        ndc_l = ops.project(verts, R.to(dev), T.to(dev), focal, S, pp)                               # light view first (:344)
        zl, _, _ = ops.depth_raster(ndc_l, mesh.topo.faces, S)
        ndc_c = ops.project(verts, cam_R.to(dev), cam_T.to(dev), focal, S, pp)                       # camera view (:351-353)
        _, face_id, ws = ops.depth_raster(ndc_c, mesh.topo.faces, S)
        lp = torch.as_tensor(self.light_posi, dtype=torch.float32).to(dev).reshape(-1, 3).expand(len(mesh), 3)
        amb = torch.as_tensor(self.amb_ratio, dtype=torch.float32).to(dev).reshape(()) * torch.ones(3, device=dev)   # :435-441
        colors = torch.cat([amb, 1.0 - amb, torch.zeros(3, device=dev)])
        return self._shade(mesh, ndc_c, face_id, ws, materials, lp, colors, S, focal, pp, zl=zl, light_R=R.to(dev), light_T=T.to(dev))


class Fragments:
    """pytorch3d.renderer.mesh.rasterizer.Fragments: what MeshRasterizer.forward returns"""

    def __init__(self, pix_to_face, zbuf, bary_coords, dists):
        self.pix_to_face, self.zbuf, self.bary_coords, self.dists = pix_to_face, zbuf, bary_coords, dists

    def __iter__(self):
        return iter((self.pix_to_face, self.zbuf, self.bary_coords, self.dists))


class MeshRasterizer:
    """MeshRasterizer(cameras=PerspectiveCameras(in_ndc=False), raster_settings=RasterizationSettings(image_size, blur_radius,
    faces_per_pixel)) (renderer_helper.py:33, 44-55, 62-79, 85-99) on the fragment-level HIP op (ops.rasterize_fragments): for callers
    that bring their own PyTorch3D-style shader.  Returns Fragments with K = faces_per_pixel slots per pixel."""

    def __init__(self, image_size, blur_radius=0.0, faces_per_pixel=1):
        self.image_size, self.blur_radius, self.faces_per_pixel = image_size, float(blur_radius), int(faces_per_pixel)

    def __call__(self, mesh, principal_point=None, focal_length=None, T=None, R=None, image_size=None, **kw):
        S = _size(image_size) if image_size is not None else self.image_size
        dev = mesh.device
        pp = _pp(principal_point) if principal_point is not None else (S / 2.0, S / 2.0)
        ndc = ops.project(mesh.verts_padded(), R.to(dev), T.to(dev), _scalar(focal_length), S, pp)
        return Fragments(*ops.rasterize_fragments(ndc, mesh.topo.faces, S, self.blur_radius, self.faces_per_pixel))


def interpolate_face_attributes(pix_to_face, bary, face_attrs):
    """pytorch3d.ops.interpolate_face_attributes: (N,H,W,K) packed face ids, (N,H,W,K,3), (F_total,3,D) -> (N,H,W,K,D), zero where empty"""
    mask = pix_to_face < 0
    a = face_attrs[pix_to_face.clamp(min=0)]
    return (bary[..., None] * a).sum(-2).masked_fill(mask[..., None], 0.0)


def softmax_rgb_blend(colors, fragments, background=(1.0, 1.0, 1.0), sigma=1e-4, gamma=1e-4, znear=1.0, zfar=100.0):
    """pytorch3d.renderer.blending.softmax_rgb_blend in torch ops: (N,H,W,K,3) -> (N,H,W,4)"""
    eps = 1e-10
    mask = (fragments.pix_to_face >= 0).to(colors.dtype)
    prob = torch.sigmoid(-fragments.dists / sigma) * mask
    alpha = torch.prod(1.0 - prob, dim=-1)
    z_inv = (zfar - fragments.zbuf) / (zfar - znear) * mask
    z_inv_max = torch.max(z_inv, dim=-1).values[..., None].clamp(min=eps)
    wnum = prob * torch.exp((z_inv - z_inv_max) / gamma)
    delta = torch.exp((eps - z_inv_max) / gamma).clamp(min=eps)
    denom = wnum.sum(-1)[..., None] + delta
    bg = torch.tensor(background, dtype=colors.dtype, device=colors.device)
    rgb = ((wnum[..., None] * colors).sum(-2) + delta * bg) / denom
    return torch.cat([rgb, (1.0 - alpha)[..., None]], -1)


def sample_textures_uv(tex, fragments, n_faces):
    """pytorch3d TexturesUV.sample_textures for K >= 1 fragments in torch ops: per-pixel uv from the packed face
    ids + barycentrics, bilinear grid_sample with align_corners=True, border padding, v axis flipped.  (N,H,W,K) -> (N,H,W,K,C)"""
    maps = tex.maps_padded()
    N, H, W, K = fragments.pix_to_face.shape
    dev = maps.device
    verts_uvs = torch.as_tensor(tex.verts_uvs, dtype=torch.float32, device=dev).reshape(-1, 2)
    faces_uvs = torch.as_tensor(tex.faces_uvs, device=dev).long().reshape(-1, 3)
    face_uv = verts_uvs[faces_uvs].repeat(N, 1, 1)                                                # packed (N*F,3,2): ids are b*F + f
    assert face_uv.shape[0] == N * n_faces
    uv = interpolate_face_attributes(fragments.pix_to_face, fragments.bary_coords, face_uv)       # (N,H,W,K,2)
    grid = torch.stack((2.0 * uv[..., 0] - 1.0, 1.0 - 2.0 * uv[..., 1]), -1)                      # flipped maps + (uv*2-1)  ==  y -> -y
    grid = grid.permute(0, 3, 1, 2, 4).reshape(N * K, H, W, 2)
    m = maps.permute(0, 3, 1, 2)[:, None].expand(N, K, -1, -1, -1).reshape(N * K, maps.shape[3], maps.shape[1], maps.shape[2])
    out = torch.nn.functional.grid_sample(m, grid, mode="bilinear", padding_mode="border", align_corners=True)
    return out.reshape(N, K, -1, H, W).permute(0, 3, 4, 1, 2)


def apply_normal_map(pixel_normals, nm):
    """PBRMaterials.apply_normal_map / compute_tangent (renderer/pbr_materials.py:58-124) in torch ops, for the fragment-level
    renderers (the K=1 shading path has it fused into csrc/shade.hip): n' = normalize(-u m.x - v m.y + n m.z)."""
    x, y, z = pixel_normals.unbind(-1)
    s = 2.0 * (z >= 0).to(z.dtype) - 1.0
    a = -1.0 / (s + z)
    b = x * y * a
    u = torch.stack((1 + s * x * x * a, s * b, -s * x), -1)
    v = torch.stack((b, s + y * y * a, -y), -1)
    out = -u * nm[..., 0:1] - v * nm[..., 1:2] + pixel_normals * nm[..., 2:3]
    return torch.nn.functional.normalize(out, dim=-1)


class NormalRenderer:
    """MeshRenderer(MeshRasterizer(K=10, blur 0), SoftPhongNormalShader) (renderer_helper.py:82-101, 192-258): the interpolated vertex
    normals (through the normal map when the materials carry one), y / z flipped, mapped to [0,1], softmax-blended over the K=10
    fragments.  Visualisation only in the reference (`vis_normal`), hence plain torch ops over the fragment-level rasteriser."""

    def __init__(self, image_size, faces_per_pixel=10):
        self.rasterizer = MeshRasterizer(image_size, 0.0, faces_per_pixel)

    def __call__(self, mesh, materials=None, **kw):
        fr = self.rasterizer(mesh, **kw)
        B = len(mesh)
        faces = mesh.topo.faces.long()
        vn = mesh.verts_normals_padded()
        fn = vn[:, faces].reshape(B * faces.shape[0], 3, 3)
        pix_n = interpolate_face_attributes(fr.pix_to_face, fr.bary_coords, fn)
        if materials is not None and getattr(materials, "use_normal_map", False):                  # renderer_helper.py:226-232 (`vis_normal`)
            pix_n = apply_normal_map(pix_n, sample_textures_uv(materials.normal_maps, fr, faces.shape[0]))
        pix_n = pix_n * torch.tensor([1.0, -1.0, -1.0], device=pix_n.device)                    # renderer_helper.py:211-212
        return softmax_rgb_blend((pix_n + 1.0) / 2.0, fr)                                          # :213, :255-257


def get_renderers(image_size, light_posi=((1.0, 1.0, -5.0),), silh_sigma=1e-7, silh_gamma=1e-1, silh_faces_per_pixel=50, device="cuda"):
    """renderer_helper.py:26-103 -> (phong_renderer, silhouette_renderer, phong_normal_renderer).  silh_gamma is inert in the
    reference too (SoftSilhouetteShader ignores it: SURVEY.md Appendix C.7)."""
    return PhongRenderer(image_size, light_posi), SilhouetteRenderer(image_size, silh_sigma, silh_faces_per_pixel), NormalRenderer(image_size, 10)


def get_shadow_renderers(image_size, light_posi=((1.0, 1.0, -5.0),), silh_sigma=1e-7, silh_gamma=1e-1, silh_faces_per_pixel=50,
                         amb_ratio=0.6, device="cuda"):
    """renderer_helper.py:416-451"""
    return MeshRendererShadow(image_size, light_posi, amb_ratio)


def process_info_for_shadow(cam, light_positions, hand_verts_center, image_size, focal_length, device="cuda"):
    """renderer_helper.py:454-468 (tiny (B,3) torch ops, differentiable)."""
    cam = cam.to(device)
    cam_T = torch.stack([-cam[:, 1], -cam[:, 2], 2 * focal_length / (image_size * cam[:, 0] + 1e-9)], dim=1)
    at_light = light_positions.to(device)
    B = cam.shape[0]
    cam_R = torch.tensor([[-1., 0., 0.], [0., -1., 0.], [0., 0., 1.]], device=device).repeat(B, 1, 1)
    radius = 1.5
    d = at_light - hand_verts_center
    pos = hand_verts_center + d * (radius / torch.linalg.norm(d, dim=1, keepdim=True))
    light_R = look_at_rotation(pos, hand_verts_center, torch.tensor([[0.0, 1.0, 0.0]], device=device))
    light_T = -torch.bmm(light_R.transpose(1, 2), pos[:, :, None])[:, :, 0]
    return light_R, light_T, cam_R, cam_T
