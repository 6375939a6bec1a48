"""torch.autograd bindings of the HIP kernels (C-ABI in include/harp_hip.h).

These are the building blocks the reference-API mirror modules (renderer/, utils/, loss/, manopth/) and the
fused engine call.  Every op takes/returns plain contiguous float32/int32 HIP tensors."""
import math

import ctypes

import numpy as np
import torch

from . import _lib

SIL_SIGMA = 1e-7                                        # optimize_sequence.py:426
SIL_BLUR = math.log(1.0 / 1e-4 - 1.0) * SIL_SIGMA       # renderer_helper.py:46


def rasterize_workspace(B, F, S, device):
    n = _lib.lib().harp_rasterize_ws_bytes(B, F, S)
    return torch.empty(n, dtype=torch.uint8, device=device)


def rasterize_ws_nact(ws, B, F, S):
    """number of (frame, 64x64 super-tile) pairs that hold faces, read back from a rasteriser workspace (csrc/harp_common.h:
    raster_ws_split — ... | nact (256 B) | hit bitmaps at the tail); debugging / tests: one D2H copy"""
    nst = ((S + 63) // 64) ** 2
    bits = B * nst * ((F + 63) // 64) * 8
    return int(ws[-(256 + bits):][:4].view(torch.int32)[0])


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


# ------------------------------------------------------------------------------------------------------
# static topology on the device
# ------------------------------------------------------------------------------------------------------
class DeviceTopology:
    """int32 HBM copies of the host tables of harp_amd.synth.build_topology / topology.py."""

    def __init__(self, topo, verts_uvs, faces_uvs, device):
        def i32(a):
            t = torch.as_tensor(np.asarray(a), dtype=torch.int32).contiguous()
            if t.numel() == 0:                             # un-subdivided template: empty tables still need a valid device pointer
                t = torch.zeros((1,) + tuple(t.shape[1:]), dtype=torch.int32)
            return t.to(device)
        self.V0, self.V = int(topo["n_verts0"]), int(topo["n_verts"])
        for k in ("faces0", "edges0", "faces", "edges", "nbr_off", "nbr_idx", "vf_off", "vf_idx", "nc_pairs", "vp_off", "vp_idx", "sub_off", "sub_idx"):
            setattr(self, k, i32(topo[k]))
        self.E0, self.F, self.E = int(np.asarray(topo["edges0"]).shape[0]), self.faces.shape[0], self.edges.shape[0]
        # expanded vertex -> incident-face table for the fused mesh chain: (i0, i1, i2, corner) per CSR entry, one 16-B load
        fc = torch.as_tensor(np.asarray(topo["vf_idx"]), dtype=torch.int64)
        fa = torch.as_tensor(np.asarray(topo["faces"]), dtype=torch.int64)
        self.vf_tri = torch.cat([fa[fc // 3], (fc % 3)[:, None]], 1).to(torch.int32).contiguous().to(device)
        self.verts_uvs = torch.as_tensor(verts_uvs, dtype=torch.float32).reshape(-1, 2).contiguous().to(device)
        self.faces_uvs = i32(faces_uvs).reshape(-1, 3)
        self.device = device
        self._check_uvs()

    def _check_uvs(self):
        # the shader kernels read verts_uvs[faces_uvs] without a bounds check
        lo, hi = int(self.faces_uvs.min()), int(self.faces_uvs.max())
        if lo < 0 or hi >= self.verts_uvs.shape[0]:
            raise ValueError(f"faces_uvs must index verts_uvs ({self.verts_uvs.shape[0]} rows): found indices in [{lo}, {hi}]")

    def set_uvs(self, verts_uvs, faces_uvs):
        """(re)bind the UV tables (TexturesUV(faces_uvs=, verts_uvs=), utils/visualize.py:84-87); accepts the reference's (1,VT,2)/(1,F,3)"""
        key = (verts_uvs.data_ptr() if torch.is_tensor(verts_uvs) else id(verts_uvs), faces_uvs.data_ptr() if torch.is_tensor(faces_uvs) else id(faces_uvs))
        if getattr(self, "_uv_key", None) != key:
            self.verts_uvs = torch.as_tensor(verts_uvs, dtype=torch.float32).reshape(-1, 2).contiguous().to(self.device)
            self.faces_uvs = torch.as_tensor(faces_uvs).to(torch.int32).reshape(-1, 3).contiguous().to(self.device)
            self._check_uvs()
            self._uv_key = key


def _f32(t):
    return t.contiguous().float()


class _Subdivide(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v0, topo, scale):
        v0 = _f32(v0)
        B = v0.shape[0]
        vs = torch.empty(B, topo.V, 3, dtype=torch.float32, device=v0.device)
        _lib.check(_lib.lib().harp_subdivide_fwd(_lib.ptr(v0), _lib.ptr(topo.edges0), B, topo.V0, topo.E0, scale, _lib.ptr(vs),
                                                 _lib.stream()), "harp_subdivide_fwd")
        ctx.topo, ctx.scale = topo, scale
        return vs

    @staticmethod
    def backward(ctx, g):
        g = _f32(g)
        topo, B = ctx.topo, g.shape[0]
        g0 = torch.empty(B, topo.V0, 3, dtype=torch.float32, device=g.device)
        _lib.check(_lib.lib().harp_subdivide_bwd(_lib.ptr(g), _lib.ptr(topo.sub_off), _lib.ptr(topo.sub_idx), B, topo.V0, topo.V,
                                                 ctx.scale, _lib.ptr(g0), _lib.stream()), "harp_subdivide_bwd")
        return g0, None, None


def subdivide(v0, topo, scale=1.0):
    """[scale*v0 ; edge midpoints] (SubdivideMeshes, utils/visualize.py:45-52)."""
    return _Subdivide.apply(v0, topo, scale)


class _NormalsDisplace(torch.autograd.Function):
    """n = unit vertex normals of v; optionally vd = v + n * disp (utils/visualize.py:58-64)."""

    @staticmethod
    def forward(ctx, v, disp, topo):
        v = _f32(v)
        B = v.shape[0]
        n = torch.empty_like(v)
        inv_len = torch.empty(B, topo.V, dtype=torch.float32, device=v.device)
        d = _f32(disp.reshape(-1)) if disp is not None else None
        vd = torch.empty_like(v) if disp is not None else None
        _lib.check(_lib.lib().harp_vertex_normals_fwd(_lib.ptr(v), _lib.ptr(topo.faces), _lib.ptr(topo.vf_off), _lib.ptr(topo.vf_idx),
                                                      B, topo.V, _lib.ptr(n), _lib.ptr(inv_len), _lib.ptr(d), _lib.ptr(vd),
                                                      _lib.stream()), "harp_vertex_normals_fwd")
        ctx.topo, ctx.has_disp = topo, disp is not None
        ctx.disp_shape = disp.shape if disp is not None else None
        ctx.save_for_backward(v, n, inv_len, d)
        return (n, vd) if disp is not None else (n, None)

    @staticmethod
    def backward(ctx, g_n, g_vd):
        v, n, inv_len, d = ctx.saved_tensors
        topo, B = ctx.topo, v.shape[0]
        L = _lib.lib()
        g_v = torch.zeros_like(v)
        g_disp = None
        g_n_tot = _f32(g_n) if g_n is not None else torch.zeros_like(v)
        if ctx.has_disp and g_vd is not None:
            g_vd = _f32(g_vd)
            g_v = g_vd.clone()
            g_nd = torch.empty_like(v)
            g_disp = torch.zeros(topo.V, dtype=torch.float32, device=v.device)
            _lib.check(L.harp_displace_bwd(_lib.ptr(g_vd), _lib.ptr(n), _lib.ptr(d), B, topo.V, _lib.ptr(g_nd), _lib.ptr(g_disp),
                                           _lib.stream()), "harp_displace_bwd")
            g_n_tot = g_n_tot + g_nd
            g_disp = g_disp.reshape(ctx.disp_shape)
        tmp = torch.empty_like(v)
        _lib.check(L.harp_vertex_normals_bwd(_lib.ptr(v), _lib.ptr(topo.faces), _lib.ptr(topo.vf_off), _lib.ptr(topo.vf_idx), B, topo.V,
                                             _lib.ptr(n), _lib.ptr(inv_len), _lib.ptr(g_n_tot.contiguous()), _lib.ptr(tmp), _lib.ptr(g_v),
                                             _lib.stream()), "harp_vertex_normals_bwd")
        return g_v, g_disp, None


def vertex_normals(v, topo):
    return _NormalsDisplace.apply(v, None, topo)[0]


def normals_displace(v, disp, topo):
    return _NormalsDisplace.apply(v, disp, topo)


class _Project(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v, R, T, focal, ppx, ppy, S):
        v, R, T = _f32(v), _f32(R).reshape(-1, 9), _f32(T)
        B, V, _ = v.shape
        ndc = torch.empty_like(v)
        _lib.check(_lib.lib().harp_project_fwd(_lib.ptr(v), _lib.ptr(R), _lib.ptr(T), B, V, focal, ppx, ppy, S, _lib.ptr(ndc),
                                               _lib.stream()), "harp_project_fwd")
        ctx.save_for_backward(v, R, T)
        ctx.meta = (focal, S)
        return ndc

    @staticmethod
    def backward(ctx, g):
        v, R, T = ctx.saved_tensors
        focal, S = ctx.meta
        B, V, _ = v.shape
        g_v = torch.zeros_like(v)
        g_R = torch.zeros(B, 9, dtype=torch.float32, device=v.device)
        g_T = torch.zeros(B, 3, dtype=torch.float32, device=v.device)
        _lib.check(_lib.lib().harp_project_bwd(_lib.ptr(v), _lib.ptr(R), _lib.ptr(T), _lib.ptr(_f32(g)), B, V, focal, S, _lib.ptr(g_v),
                                               _lib.ptr(g_R), _lib.ptr(g_T), _lib.stream()), "harp_project_bwd")
        return g_v, g_R.view(B, 3, 3), g_T, None, None, None, None


def project(v, R, T, focal, S, pp=None):
    """world -> (x_ndc, y_ndc, z_view), MeshRasterizer.transform for PerspectiveCameras(in_ndc=False)."""
    ppx, ppy = (S / 2.0, S / 2.0) if pp is None else pp
    return _Project.apply(v, R, T, float(focal), float(ppx), float(ppy), int(S))


class _DepthRaster(torch.autograd.Function):
    """K=1 hard rasterisation returning (zbuf, face_id, ws); differentiable through zbuf."""

    @staticmethod
    def forward(ctx, ndc, faces, S):
        ndc = _f32(ndc)
        face_id, zbuf, _, ws = rasterize_fwd(ndc, faces, S, soft=False, want_zbuf=True)
        ctx.save_for_backward(face_id, ws, faces)
        ctx.meta = (ndc.shape, S)
        ctx.mark_non_differentiable(face_id, ws)
        return zbuf, face_id, ws

    @staticmethod
    def backward(ctx, g_z, _a, _b):
        face_id, ws, faces = ctx.saved_tensors
        shape, S = ctx.meta
        g_ndc = torch.zeros(shape, dtype=torch.float32, device=face_id.device)
        _lib.check(_lib.lib().harp_depth_bwd(_lib.ptr(face_id), _lib.ptr(ws), _lib.ptr(faces), _lib.ptr(_f32(g_z)), shape[0], shape[1],
                                             faces.shape[0], S, _lib.ptr(g_ndc), _lib.stream()), "harp_depth_bwd")
        return g_ndc, None, None


def depth_raster(ndc, faces, S):
    return _DepthRaster.apply(ndc, faces, S)


def _shade_args(face_id, ws, topo, verts, vnormals, tex, nmap, light_pos, colors, zl, light_R, light_T, S, focal, pp, bg):
    a = _lib.ShadeArgs()
    B, V, _ = verts.shape
    for k, t in (("face_id", face_id), ("recs", ws), ("faces", topo.faces), ("faces_uvs", topo.faces_uvs), ("verts_uvs", topo.verts_uvs),
                 ("verts", verts), ("vnormals", vnormals), ("tex", tex), ("nmap", nmap), ("light_pos", light_pos), ("colors", colors),
                 ("zl", zl), ("light_R", light_R), ("light_T", light_T)):
        setattr(a, k, _lib.ptr(t))
    a.B, a.V, a.F, a.S, a.Ht, a.Wt = B, V, topo.F, S, tex.shape[-3], tex.shape[-2]
    a.focal, a.ppx, a.ppy = focal, pp[0], pp[1]
    a.bg[0], a.bg[1], a.bg[2] = bg
    return a


# the shader backward's texel gradients as records + harp_texel_reduce (include/harp_hip.h: harp_shade_args.trec) instead of the in-kernel
# scatter; False: the table form.  Same gradients either way (tests/test_gpu_parity.py::test_texel_records_match_the_table_form).
TEXEL_RECORDS = True
_trec_cache = {}


def texel_record_buffers(device, Ht, Wt, cap):
    """(records, counters, cap, double accumulators of both maps) for harp_shade_args.trec / trec_cnt / trec_cap and harp_texel_reduce /
    harp_texel_finish; None when the map has more tiles than the reduce handles"""
    nb = _lib.lib().harp_texel_bins(int(Ht), int(Wt))
    if nb > 1024:
        return None
    key = (str(device), int(Ht), int(Wt), int(cap))
    if key not in _trec_cache:
        _trec_cache.clear()
        _trec_cache[key] = (torch.empty(nb * 9 * int(cap), dtype=torch.float32, device=device),
                            torch.zeros(nb * 16 + 16, dtype=torch.int32, device=device), int(cap),
                            torch.zeros(2, int(Ht) * int(Wt) * 3, dtype=torch.float64, device=device))
    return _trec_cache[key]


class _Shade(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ndc, verts, vnormals, tex, nmap, light_pos, colors, zl, light_R, light_T, face_id, ws, topo, S, focal, pp, bg):
        verts, vnormals, tex = _f32(verts), _f32(vnormals), _f32(tex)
        nmap = _f32(nmap) if nmap is not None else None
        light_pos, colors = _f32(light_pos), _f32(colors).reshape(9)
        if zl is not None:
            zl, light_R, light_T = _f32(zl), _f32(light_R).reshape(-1, 9), _f32(light_T)
        B = verts.shape[0]
        rgb = torch.empty(B, S, S, 3, dtype=torch.float32, device=verts.device)
        a = _shade_args(face_id, ws, topo, verts, vnormals, tex, nmap, light_pos, colors, zl, light_R, light_T, S, focal, pp, bg)
        a.rgb = _lib.ptr(rgb)
        _lib.check(_lib.lib().harp_shade_fwd(a, _lib.stream()), "harp_shade_fwd")
        ctx.save_for_backward(verts, vnormals, tex, nmap, light_pos, colors, zl, light_R, light_T, face_id, ws)
        ctx.meta = (topo, S, focal, pp, bg)
        return rgb

    @staticmethod
    def backward(ctx, g_rgb):
        verts, vnormals, tex, nmap, light_pos, colors, zl, light_R, light_T, face_id, ws = ctx.saved_tensors
        topo, S, focal, pp, bg = ctx.meta
        a = _shade_args(face_id, ws, topo, verts, vnormals, tex, nmap, light_pos, colors, zl, light_R, light_T, S, focal, pp, bg)
        g_rgb = _f32(g_rgb)
        z = torch.zeros_like
        g_ndc, g_verts, g_vn, g_tex = z(verts), z(verts), z(verts), z(tex)
        g_nmap = z(nmap) if nmap is not None else None
        g_lp, g_col = z(light_pos), z(colors)
        g_zl = z(zl) if zl is not None else None
        g_lR = z(light_R) if zl is not None else None
        g_lT = z(light_T) if zl is not None else None
        for k, t in (("g_rgb", g_rgb), ("g_tex", g_tex), ("g_nmap", g_nmap), ("g_verts", g_verts), ("g_vnormals", g_vn), ("g_ndc", g_ndc),
                     ("g_zl", g_zl), ("g_light_pos", g_lp), ("g_colors", g_col), ("g_light_R", g_lR), ("g_light_T", g_lT)):
            setattr(a, k, _lib.ptr(t))
        B = verts.shape[0]
        bufs = texel_record_buffers(verts.device, a.Ht, a.Wt, max(4096, B * S * S // 8)) if TEXEL_RECORDS else None
        if bufs is not None:
            a.trec, a.trec_cnt, a.trec_cap = _lib.ptr(bufs[0]), _lib.ptr(bufs[1]), bufs[2]
            a.trec_acc_tex, a.trec_acc_nmap = _lib.ptr(bufs[3][0]), _lib.ptr(bufs[3][1])
        _lib.check(_lib.lib().harp_shade_bwd(a, _lib.stream()), "harp_shade_bwd")
        if bufs is not None:
            at, an = _lib.ptr(bufs[3][0]), (_lib.ptr(bufs[3][1]) if g_nmap is not None else None)
            _lib.check(_lib.lib().harp_texel_reduce(_lib.ptr(bufs[0]), _lib.ptr(bufs[1]), bufs[2], a.Ht, a.Wt, at, an, B * S * S // 6, _lib.stream()), "harp_texel_reduce")
            _lib.check(_lib.lib().harp_texel_finish(at, _lib.ptr(g_tex), an, _lib.ptr(g_nmap), None, a.Ht * a.Wt, _lib.stream()), "harp_texel_finish")
        return (g_ndc, g_verts, g_vn, g_tex, g_nmap, g_lp, g_col, g_zl, g_lR.view(B, 3, 3) if g_lR is not None else None, g_lT,
                None, None, None, None, None, None, None)


def shade(ndc, verts, vnormals, tex, nmap, light_pos, colors, face_id, ws, topo, S, focal, zl=None, light_R=None, light_T=None,
          pp=None, bg=(1.0, 1.0, 1.0)):
    """Fused K=1 shader (see csrc/shade.hip). `ndc` is only used to route the barycentric gradient."""
    pp = (S / 2.0, S / 2.0) if pp is None else pp
    return _Shade.apply(ndc, verts, vnormals, tex, nmap, light_pos, colors, zl, light_R, light_T, face_id, ws, topo, int(S), float(focal),
                        (float(pp[0]), float(pp[1])), tuple(float(x) for x in bg))


# ------------------------------------------------------------------------------------------------------
# fragment-level rasterisation (PyTorch3D's op pair; off the fitting loop's path)
# ------------------------------------------------------------------------------------------------------
class _RasterizeFragments(torch.autograd.Function):
    """_C.rasterize_meshes / _C.rasterize_meshes_backward for a batch sharing one face table."""

    @staticmethod
    def forward(ctx, ndc, faces, S, blur_radius, K):
        ndc = _f32(ndc)
        B, V, _ = ndc.shape
        F = faces.shape[0]
        dev = ndc.device
        ws = rasterize_workspace(B, F, S, dev)
        p2f = torch.empty(B, S, S, K, dtype=torch.int32, device=dev)
        zbuf = torch.empty(B, S, S, K, dtype=torch.float32, device=dev)
        bary = torch.empty(B, S, S, K, 3, dtype=torch.float32, device=dev)
        dists = torch.empty(B, S, S, K, dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().harp_rasterize_fragments_fwd(_lib.ptr(ndc), _lib.ptr(faces), B, V, F, S, float(blur_radius), int(K), _lib.ptr(ws),
                                                           _lib.ptr(p2f), _lib.ptr(zbuf), _lib.ptr(bary), _lib.ptr(dists), _lib.stream()),
                   "harp_rasterize_fragments_fwd")
        ctx.save_for_backward(ndc, faces, p2f)
        ctx.meta = (S, float(blur_radius), int(K))
        ctx.mark_non_differentiable(p2f)
        return p2f, zbuf, bary, dists

    @staticmethod
    def backward(ctx, _g_p2f, g_zbuf, g_bary, g_dists):
        ndc, faces, p2f = ctx.saved_tensors
        S, blur_radius, K = ctx.meta
        B, V, _ = ndc.shape
        g_ndc = torch.zeros_like(ndc)
        c = lambda g: None if g is None else _f32(g)
        gz, gb, gd = c(g_zbuf), c(g_bary), c(g_dists)
        _lib.check(_lib.lib().harp_rasterize_fragments_bwd(_lib.ptr(ndc), _lib.ptr(faces), _lib.ptr(p2f), _lib.ptr(gz), _lib.ptr(gb), _lib.ptr(gd),
                                                           B, V, faces.shape[0], S, blur_radius, K, _lib.ptr(g_ndc), _lib.stream()),
                   "harp_rasterize_fragments_bwd")
        return g_ndc, None, None, None, None


def rasterize_fragments(ndc, faces, S, blur_radius=0.0, faces_per_pixel=1, packed=True):
    """ndc (B,V,3) = (x_ndc, y_ndc, z_view), faces (F,3) int32 -> pix_to_face (B,S,S,K) int64 [PyTorch3D's packed index b*F + f when
    `packed`, -1 empty], zbuf, bary (B,S,S,K,3), dists — the outputs of pytorch3d.renderer.mesh.rasterize_meshes, differentiable
    w.r.t. ndc through zbuf / bary / dists.  faces_per_pixel caps the number of fragments kept per pixel (1..64)."""
    p2f, zbuf, bary, dists = _RasterizeFragments.apply(ndc, faces, int(S), float(blur_radius), int(faces_per_pixel))
    p2f = p2f.long()
    if packed:
        off = (torch.arange(ndc.shape[0], device=ndc.device) * faces.shape[0]).view(-1, 1, 1, 1)
        p2f = torch.where(p2f >= 0, p2f + off, p2f)
    return p2f, zbuf, bary, dists
