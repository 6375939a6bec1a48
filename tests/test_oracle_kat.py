"""Analytic known-answer tests for the PyTorch3D-semantics oracle (oracle/p3d_like.py) — the only pin
that part of the oracle has ("parity unpinned" otherwise)."""
import math

import torch

from oracle import p3d_like as P


def _tri(S=16, z=(2.0, 2.0, 2.0)):
    # right triangle in NDC covering the +x,+y quadrant corner region
    v = torch.tensor([[[0.9, 0.9, z[0]], [-0.7, 0.9, z[1]], [0.9, -0.7, z[2]]]], dtype=torch.float64)
    f = torch.tensor([[0, 1, 2]])
    return v, f


def test_pixel_centers_and_coverage():
    S = 16
    v, f = _tri(S)
    p2f, zbuf, bary, d = P.rasterize_meshes(v, f, S, 0.0, 1)
    pc = P.pixel_centers(S, torch.float64)
    assert abs(pc[0].item() - (1 - 1 / S)) < 1e-12 and abs(pc[-1].item() + (1 - 1 / S)) < 1e-12
    X, Y = torch.meshgrid(pc, pc, indexing="xy")          # [row, col] -> x of col, y of row
    inside = (X < 0.9) & (Y < 0.9) & (X + Y > 0.2)        # hypotenuse x+y = 0.2
    assert torch.equal(p2f[0, :, :, 0] >= 0, inside)
    cov = p2f[0, :, :, 0] >= 0
    assert torch.allclose(bary[0, :, :, 0][cov].sum(-1), torch.ones(int(cov.sum()), dtype=torch.float64), atol=1e-12)
    assert torch.allclose(zbuf[0, :, :, 0][cov], torch.full((int(cov.sum()),), 2.0, dtype=torch.float64))
    assert (d[0, :, :, 0][cov] < 0).all()
    # signed dist = -(squared distance to nearest edge)
    dd = torch.minimum(torch.minimum(0.9 - X, 0.9 - Y), (X + Y - 0.2) / math.sqrt(2.0)) ** 2
    assert torch.allclose(-d[0, :, :, 0][cov], dd[cov], atol=1e-12)
    assert (p2f[0, :, :, 0][~cov] == -1).all() and (zbuf[0, :, :, 0][~cov] == -1).all()


def test_perspective_correct_depth():
    S = 32
    v, f = _tri(S, z=(1.0, 3.0, 2.0))
    p2f, zbuf, bary, _ = P.rasterize_meshes(v, f, S, 0.0, 1)
    cov = p2f[0, :, :, 0] >= 0
    # 1/z is screen-linear: check against the analytic plane in (x,y,1/z)
    pc = P.pixel_centers(S, torch.float64)
    X, Y = torch.meshgrid(pc, pc, indexing="xy")
    A = torch.tensor([[0.9, 0.9, 1.0], [-0.7, 0.9, 1.0], [0.9, -0.7, 1.0]], dtype=torch.float64)
    coef = torch.linalg.solve(A, torch.tensor([1.0, 1 / 3.0, 0.5], dtype=torch.float64))
    invz = coef[0] * X + coef[1] * Y + coef[2]
    assert torch.allclose(zbuf[0, :, :, 0][cov], (1.0 / invz)[cov], atol=1e-10)


def test_blur_band_and_topk_order():
    S = 16
    v = torch.tensor([[[0.9, 0.9, 2.0], [-0.7, 0.9, 2.0], [0.9, -0.7, 2.0],
                       [0.9, 0.9, 1.0], [-0.7, 0.9, 1.0], [0.9, -0.7, 1.0]]], dtype=torch.float64)
    f = torch.tensor([[0, 1, 2], [3, 4, 5]])
    blur = 0.01
    p2f, zbuf, bary, d = P.rasterize_meshes(v, f, S, blur, 3)
    cov = p2f[0, :, :, 0] >= 0
    assert (p2f[0, :, :, 0][cov] == 1).all() and (p2f[0, :, :, 1][cov] == 0).all() and (p2f[0, :, :, 2] == -1).all()
    assert (zbuf[0, :, :, 0][cov] - 1.0).abs().max() < 1e-12 and (zbuf[0, :, :, 1][cov] - 2.0).abs().max() < 1e-12
    band = cov & (d[0, :, :, 0] > 0)
    assert band.any() and (d[0, :, :, 0][band] < blur).all()
    a = P.sigmoid_alpha_blend(p2f, d, 1e-3)
    assert ((a >= 0) & (a <= 1)).all() and (a[0][~cov] == 0).all()
    # clipped bary stays in [0,1] and sums to 1 in the band
    assert ((bary[0, :, :, 0][band] >= 0) & (bary[0, :, :, 0][band] <= 1)).all()


def test_softmax_blend_closed_form_k1():
    S = 8
    v, f = _tri(S)
    p2f, zbuf, bary, d = P.rasterize_meshes(v.float(), f, S, 0.0, 1)
    col = torch.rand(1, S, S, 1, 3)
    img = P.softmax_rgb_blend(col, p2f, zbuf, d)
    cov = p2f[0, :, :, 0] >= 0
    assert torch.allclose(img[0][cov][:, :3], col[0, :, :, 0][cov], atol=1e-6)
    assert torch.allclose(img[0][~cov][:, :3], torch.ones(int((~cov).sum()), 3))
    assert (img[0][~cov][:, 3] == 0).all() and (img[0][cov][:, 3] > 0.5).all()


def test_texture_sampling_identity_quad():
    # fronto-parallel quad with identity UVs == bilinear resample of the map (v flipped)
    S, Ht, Wt = 8, 5, 7
    v = torch.tensor([[[1.0, 1.0, 1.0], [-1.0, 1.0, 1.0], [-1.0, -1.0, 1.0], [1.0, -1.0, 1.0]]], dtype=torch.float64)
    f = torch.tensor([[0, 1, 2], [0, 2, 3]])
    uv = torch.tensor([[0.0, 1.0], [1.0, 1.0], [1.0, 0.0], [0.0, 0.0]], dtype=torch.float64)   # u grows to the right (-x ndc)
    p2f, zbuf, bary, d = P.rasterize_meshes(v, f, S, 0.0, 1)
    # pixel centres exactly on the shared diagonal have a zero barycentric -> in neither face (strict >0)
    assert torch.equal(p2f[0, :, :, 0] < 0, torch.eye(S, dtype=torch.bool))
    m = torch.rand(1, Ht, Wt, 3, dtype=torch.float64)
    tex = P.sample_textures_uv(m, uv, f, p2f, bary, 2)[0, :, :, 0]
    pc = P.pixel_centers(S, torch.float64)
    u = (1 - pc) / 2            # per column
    vv = (pc + 1) / 2           # per row
    for r in (0, 3, 7):
        for c in (1, 4, 6):
            x, y = u[c] * (Wt - 1), (1 - vv[r]) * (Ht - 1)
            x0, y0 = int(math.floor(x)), int(math.floor(y))
            x1, y1 = min(x0 + 1, Wt - 1), min(y0 + 1, Ht - 1)
            wx, wy = x - x0, y - y0
            ref = (m[0, y0, x0] * (1 - wx) * (1 - wy) + m[0, y0, x1] * wx * (1 - wy) +
                   m[0, y1, x0] * (1 - wx) * wy + m[0, y1, x1] * wx * wy)
            assert torch.allclose(tex[r, c], ref, atol=1e-10)


def test_camera_convention_and_lambert():
    S, f = 64, 300.0
    verts = torch.tensor([[[0.1, -0.2, 0.0]]], dtype=torch.float64)
    from oracle import harp_ref as H
    cam = torch.tensor([[1.5, -0.1, 0.2]], dtype=torch.float64)   # x=-c1, y=-c2 -> image centre
    R, T = H.camera_RT(cam, S, f)
    view, ndc = P.world_to_ndc(verts, R, T, f, (S / 2, S / 2), S)
    assert torch.allclose(ndc[0, 0, :2], torch.zeros(2, dtype=torch.float64), atol=1e-12)
    assert abs(ndc[0, 0, 2].item() - 2 * f / (S * 1.5 + 1e-9)) < 1e-9
    v2 = verts + torch.tensor([0.01, 0.0, 0.0], dtype=torch.float64)   # +x world -> +column (=-x ndc)
    assert P.world_to_ndc(v2, R, T, f, (S / 2, S / 2), S)[1][0, 0, 0] < 0
    xs, ys = P.view_to_screen_xy(view, f, (S / 2, S / 2), S)
    assert abs(xs.item() - S / 2) < 1e-9 and abs(ys.item() - S / 2) < 1e-9
    n = torch.tensor([[0.0, 0.0, -2.0]], dtype=torch.float64)
    p = torch.tensor([[0.0, 0.0, 1.0]], dtype=torch.float64)
    L = torch.tensor([[0.0, 1.0, 0.0]], dtype=torch.float64)
    dcol = P.point_light_diffuse(p, n, L, torch.tensor([[0.4, 0.4, 0.4]], dtype=torch.float64))
    assert torch.allclose(dcol, torch.full((1, 3), 0.4 / math.sqrt(2.0), dtype=torch.float64))
    Rl = P.look_at_rotation(torch.tensor([[0.0, 0.0, -1.5]], dtype=torch.float64), torch.zeros(1, 3, dtype=torch.float64),
                            torch.tensor([[0.0, 1.0, 0.0]], dtype=torch.float64))
    assert torch.allclose(Rl[0], torch.eye(3, dtype=torch.float64))
