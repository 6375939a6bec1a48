"""TEST INFRASTRUCTURE ONLY — CPU oracle, part 1: the PyTorch3D 0.6.2 semantics HARP's hot path uses.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this package.
The product (`harp_amd/`) never does; it fails loudly when the HIP library is missing.

PARITY UNPINNED for this file: the arithmetic restated here lives in the third-party dependency
`pytorch3d==0.6.2` (reference `requirements.txt:56`), whose source is absent from /root/reference
and which is not installable here.  What is restated is its published algorithm (SURVEY.md
Appendix A), anchored on the reference's call sites:
  rasterize_meshes            <- MeshRasterizer built at renderer/renderer_helper.py:52-55, 76-79, 444-447
  sigmoid_alpha_blend         <- SoftSilhouetteShader, renderer_helper.py:56
  softmax_rgb_blend           <- renderer_helper.py:141-143, 589-591
  interpolate_face_attributes <- renderer_helper.py:173-178, 364, 498-503
  sample_textures_uv          <- TexturesUV.sample_textures, renderer_helper.py:127, 572; pbr_materials.py:110
  verts_normals               <- Meshes.verts_normals_packed, utils/visualize.py:59; renderer_helper.py:495
  point_light_diffuse         <- _apply_lighting, renderer_helper.py:186, 513-515
  look_at_rotation            <- renderer_helper.py:466
  laplacian / normal consistency <- optimize_sequence.py:536-537
It is pinned only by the analytic known-answer tests in tests/test_oracle_kat.py.

Everything is plain torch on CPU, dtype-generic (float32 reproduces the reference's arithmetic type,
float64 is used for gradcheck-grade comparisons), differentiable through torch autograd.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

K_EPS = 1e-8  # PyTorch3D kEpsilon


# ----------------------------------------------------------------------------------------------
# Rasteriser
# ----------------------------------------------------------------------------------------------
def _edge_fn(px, py, ax, ay, bx, by):
    # EdgeFunctionForward(p, v0, v1) = (p.x-v0.x)*(v1.y-v0.y) - (p.y-v0.y)*(v1.x-v0.x)
    return (px - ax) * (by - ay) - (py - ay) * (bx - ax)


def _seg_dist2(px, py, ax, ay, bx, by):
    # PointLineDistanceForward: squared distance to the SEGMENT (a,b); degenerate edge -> |p-b|^2
    bax, bay = bx - ax, by - ay
    l2 = bax * bax + bay * bay
    safe = torch.where(l2 <= K_EPS, torch.ones_like(l2), l2)
    t = (bax * (px - ax) + bay * (py - ay)) / safe
    t = t.clamp(0.0, 1.0)
    qx, qy = ax + t * bax, ay + t * bay
    d = (px - qx) ** 2 + (py - qy) ** 2
    d_deg = (px - bx) ** 2 + (py - by) ** 2
    return torch.where(l2 <= K_EPS, d_deg, d)


def _pair_eval(fv, px, py, perspective_correct, clip_bary):
    """fv (P,3,3) face verts [x_ndc,y_ndc,z]; px,py (P,) pixel centres. Returns bary0, bary(final),
    pz, dist2 (unsigned), inside, face_area."""
    x0, y0, z0 = fv[:, 0, 0], fv[:, 0, 1], fv[:, 0, 2]
    x1, y1, z1 = fv[:, 1, 0], fv[:, 1, 1], fv[:, 1, 2]
    x2, y2, z2 = fv[:, 2, 0], fv[:, 2, 1], fv[:, 2, 2]
    # BarycentricCoordsForward: area = edge(v2, v0, v1) + eps
    area = _edge_fn(x2, y2, x0, y0, x1, y1) + K_EPS
    w0 = _edge_fn(px, py, x1, y1, x2, y2) / area
    w1 = _edge_fn(px, py, x2, y2, x0, y0) / area
    w2 = _edge_fn(px, py, x0, y0, x1, y1) / area
    if perspective_correct:
        t0, t1, t2 = w0 * z1 * z2, z0 * w1 * z2, z0 * z1 * w2
        den = (t0 + t1 + t2).clamp(min=K_EPS)
        b0, b1, b2 = t0 / den, t1 / den, t2 / den
    else:
        b0, b1, b2 = w0, w1, w2
    inside = (b0 > 0) & (b1 > 0) & (b2 > 0)
    if clip_bary:
        c0, c1, c2 = b0.clamp(0.0, 1.0), b1.clamp(0.0, 1.0), b2.clamp(0.0, 1.0)
        s = (c0 + c1 + c2).clamp(min=1e-5)
        c0, c1, c2 = c0 / s, c1 / s, c2 / s
    else:
        c0, c1, c2 = b0, b1, b2
    pz = c0 * z0 + c1 * z1 + c2 * z2
    d = torch.minimum(torch.minimum(_seg_dist2(px, py, x0, y0, x1, y1), _seg_dist2(px, py, x0, y0, x2, y2)),
                      _seg_dist2(px, py, x1, y1, x2, y2))
    return torch.stack([c0, c1, c2], 1), pz, d, inside


def pixel_centers(S, dtype):
    """NDC coordinate of pixel index i: -1 + (2*(S-1-i)+1)/S  (pixel 0 is at +1-1/S)."""
    i = torch.arange(S, dtype=dtype)
    return -1.0 + (2.0 * (S - 1 - i) + 1.0) / S


def rasterize_meshes(verts_ndc, faces, image_size, blur_radius=0.0, faces_per_pixel=1,
                     perspective_correct=True, clip_barycentric_coords=None, cull_backfaces=False, return_ambiguous=False, amb_tol=3e-7, flag_slivers=True):
    """verts_ndc (B,V,3) = (x_ndc, y_ndc, z_view); faces (F,3) long shared by the batch.

    Returns pix_to_face (B,S,S,K) int64 packed (b*F+f, -1 empty), zbuf, bary (...,3), dists (signed,
    squared NDC); the float outputs are differentiable w.r.t. verts_ndc (== rasterize_meshes_backward).

    return_ambiguous=True appends a (B,S,S) bool map of pixels whose COVERAGE by some face is not decided at float32 precision: the
    pixel centre lies within `amb_tol` (relative to the edge's extent) of an edge line while being on the inside of the other edges.
    Test infrastructure for fp32-vs-fp64 comparisons: such a pixel may legitimately land on either side in any fp32 implementation
    (including the reference's own CUDA kernels), and in a K=1 pass the flip is a discontinuity of the image.
    """
    if clip_barycentric_coords is None:
        clip_barycentric_coords = blur_radius > 0.0
    B, V, _ = verts_ndc.shape
    Fn = faces.shape[0]
    S, K = int(image_size), int(faces_per_pixel)
    dt = verts_ndc.dtype
    fv_all = verts_ndc[:, faces]                      # (B,F,3,3)
    r = math.sqrt(blur_radius)
    pc = pixel_centers(S, dt)

    with torch.no_grad():
        fvd = fv_all.detach()
        xs, ys, zs = fvd[..., 0], fvd[..., 1], fvd[..., 2]
        xmin, xmax = xs.min(-1).values, xs.max(-1).values
        ymin, ymax = ys.min(-1).values, ys.max(-1).values
        # conservative integer pixel ranges (exact float tests follow): xf(i) = 1 - (2i+1)/S
        lo_x = torch.ceil((S * (1.0 - xmax.double() - r) - 1.0) / 2.0).long() - 1
        hi_x = torch.floor((S * (1.0 - xmin.double() + r) - 1.0) / 2.0).long() + 1
        lo_y = torch.ceil((S * (1.0 - ymax.double() - r) - 1.0) / 2.0).long() - 1
        hi_y = torch.floor((S * (1.0 - ymin.double() + r) - 1.0) / 2.0).long() + 1
        lo_x, hi_x = lo_x.clamp(0, S - 1), hi_x.clamp(0, S - 1)
        lo_y, hi_y = lo_y.clamp(0, S - 1), hi_y.clamp(0, S - 1)
        offscreen = (xmax + r < -1.0) | (xmin - r > 1.0) | (ymax + r < -1.0) | (ymin - r > 1.0)
        w = (hi_x - lo_x + 1).clamp(min=0)
        h = (hi_y - lo_y + 1).clamp(min=0)
        cnt = (w * h).masked_fill(offscreen, 0).reshape(-1)          # (B*F,)
        total = int(cnt.sum())
        bf = torch.repeat_interleave(torch.arange(B * Fn), cnt)      # packed face id per pair
        start = torch.cumsum(cnt, 0) - cnt
        local = torch.arange(total) - start[bf]
        wq = w.reshape(-1)[bf]
        xi = lo_x.reshape(-1)[bf] + local % wq
        yi = lo_y.reshape(-1)[bf] + local // wq
        fvp = fvd.reshape(B * Fn, 3, 3)[bf]
        px, py = pc[xi], pc[yi]
        bary, pz, d2, inside = _pair_eval(fvp, px, py, perspective_correct, clip_barycentric_coords)
        x0, y0 = fvp[:, 0, 0], fvp[:, 0, 1]
        x1, y1 = fvp[:, 1, 0], fvp[:, 1, 1]
        x2, y2 = fvp[:, 2, 0], fvp[:, 2, 1]
        face_area = _edge_fn(x0, y0, x1, y1, x2, y2)
        zero_area = (face_area <= K_EPS) & (face_area >= -K_EPS)
        zmax = fvp[:, :, 2].max(1).values
        zmin = fvp[:, :, 2].min(1).values
        out_bbox = (px > xmax.reshape(-1)[bf] + r) | (px < xmin.reshape(-1)[bf] - r) | \
                   (py > ymax.reshape(-1)[bf] + r) | (py < ymin.reshape(-1)[bf] - r) | (zmin < K_EPS)
        ok = ~((zmax < 0) | out_bbox | zero_area)
        amb_map = None
        if return_ambiguous:
            sgn = torch.sign(face_area)
            near_all, in_all = torch.zeros_like(ok), torch.ones_like(ok)
            for (ax, ay, bx, by) in ((x1, y1, x2, y2), (x2, y2, x0, y0), (x0, y0, x1, y1)):
                e = _edge_fn(px, py, ax, ay, bx, by)
                near = e.abs() < amb_tol * ((bx - ax).abs() + (by - ay).abs() + (px - ax).abs() + (py - ay).abs())
                near_all |= near
                in_all &= near | (e * sgn > 0)
            flag = ok & near_all & in_all
            amb_map = torch.zeros(B * S * S, dtype=torch.bool)
            amb_map[((bf // Fn) * (S * S) + yi * S + xi)[flag]] = True
            amb_map = amb_map.view(B, S, S)
        if cull_backfaces:
            ok &= ~(face_area < 0)
        ok &= ~(pz < 0)
        ok &= ~((~inside) & (d2 >= blur_radius))
        bf, xi, yi, pz = bf[ok], xi[ok], yi[ok], pz[ok]
        pix = (bf // Fn) * (S * S) + yi * S + xi
        order = np.lexsort((bf.numpy(), pz.numpy(), pix.numpy()))   # by pixel, then z, then face id
        order = torch.from_numpy(order)
        pix_s, bf_s = pix[order], bf[order]
        first = torch.ones_like(pix_s, dtype=torch.bool)
        first[1:] = pix_s[1:] != pix_s[:-1]
        gstart = torch.cummax(torch.where(first, torch.arange(len(pix_s)), torch.zeros_like(pix_s)), 0).values
        rank = torch.arange(len(pix_s)) - gstart
        keep = rank < K
        pix_k, bf_k, rank_k = pix_s[keep], bf_s[keep], rank[keep]
        slot = pix_k * K + rank_k

    pix_to_face = torch.full((B * S * S * K,), -1, dtype=torch.int64)
    pix_to_face[slot] = bf_k
    # differentiable recomputation on the selected (pixel, face) pairs
    fvs = fv_all.reshape(B * Fn, 3, 3)[bf_k]
    rem = pix_k % (S * S)
    px, py = pc[rem % S], pc[rem // S]
    bary, pz, d2, inside = _pair_eval(fvs, px, py, perspective_correct, clip_barycentric_coords)
    sd = torch.where(inside, -d2, d2)
    zbuf = torch.full((B * S * S * K,), -1.0, dtype=dt).index_put((slot,), pz)
    dists = torch.full((B * S * S * K,), -1.0, dtype=dt).index_put((slot,), sd)
    baryo = torch.full((B * S * S * K, 3), -1.0, dtype=dt).index_put((slot,), bary)
    shp = (B, S, S, K)
    if return_ambiguous:
        # ... and pixels whose nearest face is a SLIVER in NDC: its barycentric gradients scale with 1/area, and the area of a triangle
        # whose vertices are float32 numbers of magnitude ~0.5 (absolute uncertainty ~4e-8 each) is only known to ~4e-8 * perimeter
        with torch.no_grad():
            f2 = fvd.reshape(B * Fn, 3, 3)
            ex = (f2[:, 1, :2] - f2[:, 0, :2]).abs().sum(-1) + (f2[:, 2, :2] - f2[:, 0, :2]).abs().sum(-1) + (f2[:, 2, :2] - f2[:, 1, :2]).abs().sum(-1)
            area = _edge_fn(f2[:, 0, 0], f2[:, 0, 1], f2[:, 1, 0], f2[:, 1, 1], f2[:, 2, 0], f2[:, 2, 1]).abs()
            sliver = (4e-8 * ex) > 3e-4 * area
            p0 = pix_to_face.view(shp)[..., 0]
            if flag_slivers:
                amb_map = amb_map | ((p0 >= 0) & sliver[p0.clamp(min=0)])
        return pix_to_face.view(shp), zbuf.view(shp), baryo.view(*shp, 3), dists.view(shp), amb_map
    return pix_to_face.view(shp), zbuf.view(shp), baryo.view(*shp, 3), dists.view(shp)


# ----------------------------------------------------------------------------------------------
# Fragment consumers
# ----------------------------------------------------------------------------------------------
def interpolate_face_attributes(pix_to_face, bary, face_attrs):
    """face_attrs (F_total,3,D) -> (N,H,W,K,D); zeros where pix_to_face < 0 (Appendix A.5)."""
    mask = pix_to_face < 0
    idx = pix_to_face.clamp(min=0)
    a = face_attrs[idx]                               # (N,H,W,K,3,D)
    out = (bary[..., None] * a).sum(-2)
    return out.masked_fill(mask[..., None], 0.0)


def sigmoid_alpha_blend(pix_to_face, dists, sigma):
    """alpha channel of SoftSilhouetteShader (Appendix A.3): 1 - prod_k (1 - sigmoid(-d/sigma)*mask)."""
    mask = (pix_to_face >= 0).to(dists.dtype)
    prob = torch.sigmoid(-dists / sigma) * mask
    return 1.0 - torch.prod(1.0 - prob, dim=-1)


def softmax_rgb_blend(colors, pix_to_face, zbuf, dists, sigma=1e-4, gamma=1e-4,
                      background=(1.0, 1.0, 1.0), znear=1.0, zfar=100.0):
    """(N,H,W,K,3) colours -> (N,H,W,4) (Appendix A.4)."""
    eps = 1e-10
    dt = colors.dtype
    mask = (pix_to_face >= 0).to(dt)
    prob = torch.sigmoid(-dists / sigma) * mask
    alpha = torch.prod(1.0 - prob, dim=-1)
    z_inv = (zfar - zbuf) / (zfar - znear) * mask
    z_inv_max = torch.max(z_inv, dim=-1).values[..., None].clamp(min=eps)
    wnum = prob * torch.exp((z_inv - z_inv_max) / gamma)
    delta = torch.exp((eps - z_inv_max) / gamma).clamp(min=eps)
    denom = wnum.sum(-1)[..., None] + delta
    bg = torch.tensor(background, dtype=dt)
    rgb = ((wnum[..., None] * colors).sum(-2) + delta * bg) / denom
    return torch.cat([rgb, (1.0 - alpha)[..., None]], -1)


def sample_textures_uv(maps, verts_uvs, faces_uvs, pix_to_face, bary, faces_per_mesh):
    """TexturesUV.sample_textures (Appendix A.6). maps (N,Ht,Wt,C); verts_uvs (VT,2); faces_uvs (F,3)
    shared by the batch; returns (N,H,W,K,C)."""
    N, H, W, Kk = pix_to_face.shape
    fuv = verts_uvs[faces_uvs]                                    # (F,3,2)
    fuv = fuv.repeat(N, 1, 1)                                     # packed over the batch
    puv = interpolate_face_attributes(pix_to_face, bary, fuv)     # (N,H,W,K,2)
    puv = puv.permute(0, 3, 1, 2, 4).reshape(N * Kk, H, W, 2)
    C = maps.shape[-1]
    tm = maps.permute(0, 3, 1, 2)[None].expand(Kk, -1, -1, -1, -1).transpose(0, 1).reshape(N * Kk, C, *maps.shape[1:3])
    lo = puv.new_tensor([-1.0, 1.0])
    hi = puv.new_tensor([1.0, -1.0])
    grid = torch.lerp(lo, hi, puv)                                # x: 2u-1, y: 1-2v
    tex = F.grid_sample(tm, grid, mode="bilinear", align_corners=True, padding_mode="border")
    return tex.reshape(N, Kk, C, H, W).permute(0, 3, 4, 1, 2)


# ----------------------------------------------------------------------------------------------
# Mesh helpers
# ----------------------------------------------------------------------------------------------
def verts_normals(verts, faces):
    """(B,V,3),(F,3) -> unit area-weighted vertex normals (Appendix A.7)."""
    fv = verts[:, faces]                                          # (B,F,3,3)
    fn = torch.cross(fv[:, :, 2] - fv[:, :, 1], fv[:, :, 0] - fv[:, :, 1], dim=-1)
    vn = torch.zeros_like(verts)
    for k in range(3):
        vn = vn.index_add(1, faces[:, k], fn)
    return F.normalize(vn, eps=1e-6, dim=-1)


def look_at_rotation(camera_position, at, up):
    """Appendix A.9. (B,3) each -> (B,3,3) with columns x,y,z."""
    z = F.normalize(at - camera_position, eps=1e-5)
    x = F.normalize(torch.cross(up.expand_as(z), z, dim=1), eps=1e-5)
    y = F.normalize(torch.cross(z, x, dim=1), eps=1e-5)
    close = torch.isclose(x, torch.zeros((), dtype=x.dtype), atol=5e-3).all(dim=1, keepdim=True)
    if close.any():
        x = torch.where(close, F.normalize(torch.cross(y, z, dim=1), eps=1e-5), x)
    R = torch.cat((x[:, None, :], y[:, None, :], z[:, None, :]), dim=1)
    return R.transpose(1, 2)


def world_to_ndc(verts, R, T, focal, pp, S):
    """MeshRasterizer.transform for PerspectiveCameras(in_ndc=False) (Appendix A.1).
    verts (B,V,3); R (B,3,3); T (B,3). Returns (view (B,V,3), ndc (B,V,3) with z = view z)."""
    view = torch.bmm(verts, R) + T[:, None, :]
    z = view[..., 2]
    xs = focal * view[..., 0] / z + pp[0]
    ys = focal * view[..., 1] / z + pp[1]
    half = S / 2.0
    x_ndc = (xs - 2.0 * pp[0] + half) / half
    y_ndc = (ys - 2.0 * pp[1] + half) / half
    return view, torch.stack([x_ndc, y_ndc, z], -1)


def view_to_screen_xy(view, focal, pp, S):
    """cameras.transform_points_screen (x right, y down, pixel units) from view-space points."""
    z = view[..., 2]
    half = S / 2.0
    x_ndc = (focal * view[..., 0] / z - pp[0] + half) / half
    y_ndc = (focal * view[..., 1] / z - pp[1] + half) / half
    return half - half * x_ndc, half - half * y_ndc


def point_light_diffuse(points, normals, light_location, diffuse_color):
    """PointLights.diffuse (Appendix A.8): colour * relu(n^ . l^); eps 1e-6 normalisations."""
    n = F.normalize(normals, p=2, dim=-1, eps=1e-6)
    d = F.normalize(light_location - points, p=2, dim=-1, eps=1e-6)
    ang = F.relu((n * d).sum(-1))
    return diffuse_color * ang[..., None]


def mesh_laplacian_smoothing_uniform(verts, nbr_off, nbr_idx):
    """Appendix A.11 with a CSR neighbour table; verts (B,V,3) -> scalar."""
    B, V, _ = verts.shape
    deg = (nbr_off[1:] - nbr_off[:-1]).to(verts.dtype)
    row = torch.repeat_interleave(torch.arange(V), (nbr_off[1:] - nbr_off[:-1]).long())
    acc = torch.zeros_like(verts).index_add(1, row, verts[:, nbr_idx.long()])
    lv = acc / deg[None, :, None] - verts
    return (lv.norm(dim=-1) / V).sum() / B


def mesh_normal_consistency(verts, pairs):
    """Appendix A.12; pairs (P,4) = [v0, v1, a, b]."""
    B = verts.shape[0]
    p = pairs.long()
    v0, v1, a, b = verts[:, p[:, 0]], verts[:, p[:, 1]], verts[:, p[:, 2]], verts[:, p[:, 3]]
    n0 = torch.cross(v1 - v0, a - v0, dim=-1)
    n1 = -torch.cross(v1 - v0, b - v0, dim=-1)
    # torch 1.11 (the reference's pinned version, requirements.txt:81) cosine_similarity:
    #   w12 / sqrt(clamp_min(w1 * w2, eps^2)), eps = 1e-8 — the PRODUCT of the squared norms is clamped, which is active
    #   for millimetre-sized triangles in metre units (|n0||n1| ~ 1e-11): the term is then ~1 with a tiny gradient.
    #   (torch >= 1.12 clamps each norm separately; restating 1.11 keeps the reference's behaviour.)
    w12, w1, w2 = (n0 * n1).sum(-1), (n0 * n0).sum(-1), (n1 * n1).sum(-1)
    loss = 1.0 - w12 / (w1 * w2).clamp_min(1e-8 * 1e-8).sqrt()
    return (loss / p.shape[0]).sum() / B
