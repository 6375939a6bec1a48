"""-m gpu: the fragment-level rasteriser (harp_rasterize_fragments_fwd/bwd = PyTorch3D's rasterize_meshes / rasterize_meshes_backward,
SURVEY.md §8b) against oracle/p3d_like.rasterize_meshes: K=1 and K=10 hard passes (renderer_helper.py:76-101), the K=50 blurred
silhouette pass (:44-58), a binding cap (K=2), gradients through zbuf / bary / dists, and the K=10 normal renderer built on it."""
import numpy as np
import pytest
import torch

from tests._scene import make_scene, rel

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def geom():
    from oracle import harp_ref as H, p3d_like as P
    sc = make_scene(T=2, S=96, seed=5)
    S, focal = 96, sc["focal"]
    params = dict(pose=sc["seq"]["pose"], rot=sc["seq"]["rot"], trans=sc["seq"]["trans"], shape=sc["seq"]["shape"].mean(0), verts_disps=torch.zeros(3093, 1))
    fid = torch.arange(2)
    with torch.no_grad():
        _, v = H.prepare_mesh(params, fid, sc["model"], sc["topo"])
        R, T = H.camera_RT(sc["seq"]["cam"][fid], S, focal)
        _, ndc = P.world_to_ndc(v.double(), R.double(), T.double(), focal, (S / 2, S / 2), S)
    return sc, ndc.float(), v, S, focal


@pytest.mark.parametrize("K,blur", [(1, 0.0), (10, 0.0), (2, 0.0), (50, None), (3, 2e-4)])
def test_fragments_match_oracle(geom, K, blur):
    from harp_amd import ops
    from oracle import p3d_like as P
    sc, ndc, _, S, _ = geom
    blur = ops.SIL_BLUR if blur is None else blur
    faces = sc["topo"]["faces"]
    Fn = faces.shape[0]
    x = ndc.clone().double().requires_grad_()          # oracle in float64 on the float32-rounded NDC vertices
    p2f_o, z_o, b_o, d_o = P.rasterize_meshes(x, faces, S, blur, K)
    xd = ndc.clone().to(DEV).requires_grad_()
    p2f, z, b, d = ops.rasterize_fragments(xd, faces.int().to(DEV), S, blur, K)
    assert p2f.shape == (2, S, S, K) and p2f.dtype == torch.int64
    same = p2f.cpu() == p2f_o
    assert (~same).float().mean() < 2e-4                # packed ids (b*F + f), identical except depth near-ties / on-edge pixels
    assert int(p2f.max()) < 2 * Fn and int(p2f[1][p2f[1] >= 0].min()) >= Fn
    m = same & (p2f_o >= 0)
    assert (z.cpu().double() - z_o)[m].abs().max() < 2e-6
    db = (b.cpu().double() - b_o)[m].abs()
    assert db.max() < 2e-3 and (db > 1e-4).double().mean() < 1e-3      # (sub-pixel faces at S = 96: 1/area amplifies float32 rounding)
    assert (d.cpu().double() - d_o)[m].abs().max() < 1e-7
    empty = p2f_o < 0
    assert (z.cpu()[empty & same] == -1).all() and (d.cpu()[empty & same] == -1).all() and (b.cpu()[empty & same] == -1).all()
    if K > 1:                                           # ascending depth per pixel
        zz = torch.where(p2f.cpu() >= 0, z.cpu(), torch.full_like(z.cpu(), 1e9))
        assert (zz[..., 1:] >= zz[..., :-1]).all()
    # gradients through all three float outputs (random cotangents on the slots both sides agree on)
    g = torch.Generator().manual_seed(K)
    wz, wb, wd = torch.randn(z_o.shape, generator=g, dtype=torch.float64), torch.randn(b_o.shape, generator=g, dtype=torch.float64), torch.randn(d_o.shape, generator=g, dtype=torch.float64)
    mm = m.double()
    (((z_o * wz + d_o * wd * 1e3) * mm).sum() + (b_o * wb * mm[..., None]).sum()).backward()
    mmd = mm.float().to(DEV)
    (((z * wz.float().to(DEV) + d * wd.float().to(DEV) * 1e3) * mmd).sum() + (b * wb.float().to(DEV) * mmd[..., None]).sum()).backward()
    assert rel(xd.grad.cpu().double(), x.grad) < 2e-3, rel(xd.grad.cpu().double(), x.grad)


def test_normal_renderer_and_capped_silhouette(geom):
    """get_renderers()[2] (K=10 normal renderer, renderer_helper.py:82-101) and a silhouette renderer whose faces_per_pixel cap binds"""
    from harp_amd import ops
    from harp_amd.renderer import renderer_helper as RH
    from harp_amd.structures import Meshes
    from harp_amd.utils.visualize import MeshSubdivider
    from oracle import harp_ref as H, p3d_like as P
    sc, ndc, v, S, focal = geom
    sub = MeshSubdivider(torch.from_numpy(sc["tpl"]["faces0"]), 778, DEV)
    mesh = Meshes(v.float().to(DEV), sub.faces, None, sub.topo)
    cam = sc["seq"]["cam"][:2]
    R, T = H.camera_RT(cam, S, focal)
    kw = dict(principal_point=torch.Tensor([(S / 2., S / 2.)]), focal_length=focal, T=T.to(DEV), R=R.to(DEV), image_size=torch.Tensor([(S, S)]))
    _, _, normal_renderer = RH.get_renderers(image_size=S, device=DEV)
    img = normal_renderer(mesh, **kw)
    assert img.shape == (2, S, S, 4)
    faces = sc["topo"]["faces"]
    p2f, z, b, d = P.rasterize_meshes(ndc.double(), faces, S, 0.0, 10)
    vn = P.verts_normals(v.double(), faces)
    pn = P.interpolate_face_attributes(p2f, b, vn[:, faces].reshape(-1, 3, 3)) * torch.tensor([1.0, -1.0, -1.0], dtype=torch.float64)
    ref = P.softmax_rgb_blend((pn + 1.0) / 2.0, p2f, z, d)
    # (gamma = 1e-4 on z_inv = (100 - z)/99 turns a float32 depth error of 1e-7 m into 1e-3 of blending weight wherever two of the 10
    #  fragments are millimetres apart: the comparison is against the float64 oracle)
    dimg = (img.cpu().double() - ref).abs().max(-1).values
    assert dimg.mean() < 1e-4 and (dimg > 5e-3).float().mean() < 5e-3
    # K = 2 silhouette: only the two nearest fragments enter the product (the fused kernel would use every face in the blur band)
    sil = RH.SilhouetteRenderer(S, 1e-7, 2)(mesh, **kw)[..., 3]
    p2, _, _, d2 = P.rasterize_meshes(ndc.double(), faces, S, ops.SIL_BLUR, 2)
    a_ref = P.sigmoid_alpha_blend(p2, d2, 1e-7)
    assert ((sil.cpu().double() - a_ref).abs() > 1e-4).float().mean() < 1e-3


def test_normal_renderer_through_the_normal_map(geom):
    """`vis_normal` with materials that carry a normal map (renderer_helper.py:226-232): K=10 fragments -> TexturesUV.sample_textures ->
    PBRMaterials.apply_normal_map -> flip / [0,1] map -> softmax blend, against the oracle's restatement of the same chain"""
    from harp_amd.renderer import renderer_helper as RH
    from harp_amd.renderer.pbr_materials import PBRMaterials
    from harp_amd.structures import Meshes, TexturesUV
    from harp_amd.utils.visualize import MeshSubdivider
    from oracle import harp_ref as H, p3d_like as P
    sc, ndc, v, S, focal = geom
    sub = MeshSubdivider(torch.from_numpy(sc["tpl"]["faces0"]), 778, DEV)
    mesh = Meshes(v.float().to(DEV), sub.faces, None, sub.topo)
    R, T = H.camera_RT(sc["seq"]["cam"][:2], S, focal)
    g = torch.Generator().manual_seed(9)
    nmap = torch.nn.functional.normalize(torch.tensor([0., 0., 1.]).repeat(1, 512, 512, 1) + torch.randn(1, 512, 512, 3, generator=g) * 0.2, dim=-1)
    vuv, fuv = torch.from_numpy(sc["tpl"]["verts_uvs"]), torch.from_numpy(sc["tpl"]["faces_uvs"]).long()
    mats = PBRMaterials(shininess=0.0, normal_maps=TexturesUV(maps=nmap.repeat(2, 1, 1, 1).to(DEV), faces_uvs=fuv, verts_uvs=vuv))
    kw = dict(principal_point=torch.Tensor([(S / 2., S / 2.)]), focal_length=focal, T=T.to(DEV), R=R.to(DEV), image_size=torch.Tensor([(S, S)]))
    img = RH.NormalRenderer(S, 10)(mesh, materials=mats, **kw)
    faces = sc["topo"]["faces"]
    p2f, z, b, d = P.rasterize_meshes(ndc.double(), faces, S, 0.0, 10)
    vn = P.verts_normals(v.double(), faces)
    pn = P.interpolate_face_attributes(p2f, b, vn[:, faces].reshape(-1, 3, 3))
    nm = P.sample_textures_uv(nmap.double().repeat(2, 1, 1, 1), vuv.double(), fuv, p2f, b, faces.shape[0])
    pn = H.apply_normal_map(pn, nm) * torch.tensor([1.0, -1.0, -1.0], dtype=torch.float64)
    ref = P.softmax_rgb_blend((pn + 1.0) / 2.0, p2f, z, d)
    dimg = (img.cpu().double() - ref).abs().max(-1).values
    assert dimg.mean() < 2e-4 and (dimg > 5e-3).float().mean() < 1e-2, (dimg.mean().item(), (dimg > 5e-3).float().mean().item())
    assert (img[..., :3].cpu() - RH.NormalRenderer(S, 10)(mesh, **kw)[..., :3].cpu()).abs().max() > 0.05      # the map really perturbs the normals


@pytest.mark.parametrize("kind,S", [("hand", 512), ("arm", 1024)])
def test_depth_complexity_stays_below_the_silhouette_cap(kind, S):
    """The fused soft-silhouette kernel lets EVERY face within the blur radius of a pixel contribute, the reference keeps the 50 nearest
    (renderer_helper.py:52-55).  Counting the fragments per pixel of the bench scenes (same generator as bench.py) with the
    fragment-level op (64 slots): the count stays far below 50, so the cap cannot bind and both give the same alpha."""
    import bench
    from harp_amd import ops
    eng, _ = bench.build_engine(0, 1, torch.device(DEV), T=8, img=S, B=8, kind=kind)
    eng.fid.copy_(torch.arange(8, dtype=torch.int32, device=DEV)); eng.tfid.zero_()
    eng.forward_backward(True, True)
    torch.cuda.synchronize()
    worst = 0
    for b in range(8):
        p2f, _, _, _ = ops.rasterize_fragments(eng.s["ndc_c"][b:b + 1].clone(), eng.topo.faces, S, ops.SIL_BLUR, 64)
        worst = max(worst, int((p2f >= 0).sum(-1).max()))
    print(f"[depth complexity] {kind} {S}x{S}: at most {worst} faces within the blur radius of one pixel (cap 50)")
    assert 2 <= worst < 40, worst
