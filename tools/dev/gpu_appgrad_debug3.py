"""isolate the shader backward: leaf ndc / verts / vertex normals fed to both the HIP shader (ops.shade, autograd) and the fp64 oracle pieces"""
import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, numpy as np
import torch.nn.functional as F
from harp_amd import ops
from tests._scene import make_fit_case, mask_ambiguous_pixels, oracle_inputs, rel
from oracle import harp_ref as H, p3d_like as P

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 3
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
case = make_fit_case("hand", T=2, S=S, B=2, seed=seed, device="cuda", self_shadow=False)
eng = case["eng"]
mask_ambiguous_pixels(case)
Pm, model, tg = oracle_inputs(case)
fid = torch.arange(2)
focal = case["focal"]
with torch.no_grad():
    _, v0 = H.prepare_mesh(Pm, fid, model, case["topo"])
    faces = case["topo"]["faces"]
    vn0 = P.verts_normals(v0, faces)
    R, T = H.camera_RT(Pm["cam"][fid], S, focal)
    _, ndc0 = P.world_to_ndc(v0, R, T, focal, (S / 2, S / 2), S)
tex0 = Pm["texture"].detach()
nm0 = F.normalize(Pm["normal_map"].detach(), dim=-1)
lp = Pm["light_positions"][0].detach().repeat(2, 1)
y_true, y_col = tg["y_true"][fid], tg["y_sil_col"][fid]

def oracle(ndc, v, vn, tex, nm):
    Fn = faces.shape[0]
    p2f, zbuf, bary, dists = P.rasterize_meshes(ndc, faces, S, 0.0, 1)
    fverts, fnorm = v[:, faces].reshape(2 * Fn, 3, 3), vn[:, faces].reshape(2 * Fn, 3, 3)
    pix_pos = P.interpolate_face_attributes(p2f, bary, fverts)
    texels = P.sample_textures_uv(tex.repeat(2, 1, 1, 1), Pm["verts_uvs"], Pm["faces_uvs"], p2f, bary, Fn)
    pix_n = P.interpolate_face_attributes(p2f, bary, fnorm)
    nmm = P.sample_textures_uv(nm.repeat(2, 1, 1, 1), Pm["verts_uvs"], Pm["faces_uvs"], p2f, bary, Fn)
    pix_n = H.apply_normal_map(pix_n, nmm)
    dc = torch.full((1, 3), 0.4, dtype=v.dtype)
    diff = P.point_light_diffuse(pix_pos, pix_n, lp[:, None, None, None, :], dc[:, None, None, None, :])
    colors = (0.5 + diff) * texels + 0.1
    img = P.softmax_rgb_blend(colors, p2f, zbuf, dists)[..., :3]
    return F.l1_loss(y_true * y_col.unsqueeze(-1), img * y_col.unsqueeze(-1)), p2f

leaf = [t.clone().requires_grad_() for t in (ndc0, v0, vn0, tex0, nm0)]
loss_o, p2f = oracle(*leaf)
loss_o.backward()

dev = "cuda"
topo = eng.topo
hl = [t.detach().float().to(dev).requires_grad_() for t in (ndc0, v0, vn0, tex0[0], nm0[0])]
_, face_id, ws = ops.depth_raster(hl[0], topo.faces, S)
colors = torch.tensor([0.5] * 3 + [0.4] * 3 + [0.1] * 3, device=dev)
rgb = ops.shade(hl[0], hl[1], hl[2], hl[3], hl[4], lp.float().to(dev), colors, face_id, ws, topo, S, focal)
yt, yc = y_true.float().to(dev), y_col.float().to(dev)
loss_h = F.l1_loss(yt * yc.unsqueeze(-1), rgb * yc.unsqueeze(-1))
loss_h.backward()
print("loss", loss_h.item(), loss_o.item())
fo = torch.where(p2f[..., 0] >= 0, p2f[..., 0] % faces.shape[0], p2f[..., 0])
print("face id mismatches", (face_id.cpu() != fo).sum().item())
names = ["ndc", "verts", "vnormals", "texture", "nmap"]
for n, a, b in zip(names, hl, leaf):
    ga, gb = a.grad.cpu().double().reshape(b.grad.shape), b.grad
    print(f"{n:9s} rel {rel(ga, gb):.2e}  norm {gb.norm().item():.3e}")
    if n in ("ndc", "verts", "vnormals"):
        for c in range(3):
            print(f"    comp {c}: rel {rel(ga[..., c], gb[..., c]):.2e} norm {gb[..., c].norm().item():.3e}")
        d = (ga - gb).norm(dim=-1)
        top = torch.topk(d.flatten(), 6)
        for val, i in zip(top.values.tolist(), top.indices.tolist()):
            bb, vv = divmod(i, ga.shape[1])
            print(f"      vertex ({bb},{vv}) |diff| {val:.3e} hip {[f'{t:.3e}' for t in ga[bb, vv].tolist()]} ref {[f'{t:.3e}' for t in gb[bb, vv].tolist()]}")
