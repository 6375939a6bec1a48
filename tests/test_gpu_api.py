"""-m gpu: the reference-API mirror (prepare_mesh / get_renderers / render_image[_with_RT] / losses, driven by torch autograd exactly
like the loop body optimize_sequence.py:446-569) against the CPU oracle; checkpoint round trip; the fitting entry point."""
import numpy as np
import pytest
import torch

from tests._scene import check_removed, make_scene, mask_scene_targets, oracle_params, rel

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_loop_body_through_reference_api():
    from harp_amd.loss.arap import arap_loss, mesh_laplacian_smoothing, mesh_normal_consistency
    from harp_amd.loss.kps_loss import kps_loss
    from harp_amd.loss.texture_reg import albedo_reg, normal_reg
    from harp_amd.manopth.manolayer import ManoLayer
    from harp_amd.optimize_sequence import get_mesh_subdivider, init_params
    from harp_amd.renderer import renderer_helper
    from harp_amd.structures import Meshes
    from harp_amd.utils.visualize import prepare_materials, prepare_mesh, render_image, render_image_with_RT
    from oracle import harp_ref as H
    sc = make_scene(T=3, S=128, seed=1)
    S, focal, tg = sc["S"], sc["focal"], sc["targets"]
    configs = dict(model_type="harp", img_size=S, focal_length=focal, use_arm=False, self_shadow=True, share_light_position=True)
    layer = ManoLayer(mano_root="unused", flat_hand_mean=False, use_pca=False, model=sc["model_np"], device=DEV)
    sub = get_mesh_subdivider(layer, use_arm=False, device=DEV)
    params = init_params(sc["seq"], True, True, None, layer.th_faces, False, torch.from_numpy(sc["tpl"]["verts_uvs"])[None],
                         torch.from_numpy(sc["tpl"]["faces_uvs"])[None], configs=configs, device=DEV, uv_mask=sc["uv_mask"])
    with torch.no_grad():
        params["verts_disps"].copy_(torch.randn(3093, 1) * 0.001)
        params["texture"].copy_(torch.rand(1, 512, 512, 3) * 0.5 + 0.3)
        params["trans"].copy_(torch.randn(3, 3) * 0.01)      # all-zero trans takes the reference's no-translation branch (manolayer.py:281)
    fid = torch.tensor([1, 2])
    B = 2
    # pixels whose colour is not decided at float32 precision (tests/_scene.mask_ambiguous_pixels) leave the photometric mask of both paths
    tg, removed = mask_scene_targets(sc, params, fid)
    check_removed("api_loop_body_128", removed)
    P = oracle_params(sc, params)
    # ---- the loop body, reference call for call (optimize_sequence.py:453-553)
    with torch.no_grad():
        _, rverts, rfaces, rtex = prepare_mesh(params, torch.tensor([0]), layer, False, sub, False, configs, device=DEV)
        ref_meshes = Meshes(rverts, rfaces, rtex)
    light_positions = params["light_positions"][0].repeat(B, 1)
    phong_renderer, silhouette_renderer, _ = renderer_helper.get_renderers(image_size=S, light_posi=light_positions, silh_sigma=1e-7,
                                                                          silh_gamma=1e-1, silh_faces_per_pixel=50, device=DEV)
    hand_joints, hand_verts, faces, textures = prepare_mesh(params, fid, layer, False, sub, False, configs, device=DEV)
    meshes = Meshes(hand_verts, faces, textures)
    cam = params["cam"][fid.to(DEV)]
    materials_properties = prepare_materials(params, B, device=DEV)
    y_sil_pred = render_image(meshes, cam, B, silhouette_renderer, S, focal, silhouette=True, device=DEV)
    light_R, light_T, cam_R, cam_T = renderer_helper.process_info_for_shadow(cam, light_positions, hand_verts.mean(1), image_size=S,
                                                                             focal_length=focal, device=DEV)
    shadow_renderer = renderer_helper.get_shadow_renderers(image_size=S, light_posi=light_positions, amb_ratio=torch.sigmoid(params["amb_ratio"]),
                                                           device=DEV)
    y_pred = render_image_with_RT(meshes, light_T, light_R, cam_T, cam_R, B, shadow_renderer, S, focal, materials_properties=materials_properties,
                                  device=DEV)
    y_true, y_sil, y_col = (tg[k][fid].to(DEV) for k in ("y_true", "y_sil", "y_sil_col"))
    l1 = torch.nn.L1Loss()
    g = torch.Generator().manual_seed(3)
    d_a = torch.normal(0, 1.0, (512, 512, 2), generator=g).to(torch.int)
    d_n = torch.normal(0, 2.0, (512, 512, 2), generator=g).to(torch.int)
    loss = {"silhouette": l1(y_sil, y_sil_pred),
            "kps_anchor": kps_loss(params["init_joints"][fid], hand_joints, use_arm=False),
            "vert_disp_reg": torch.sum(params["verts_disps"] ** 2.0),
            "laplacian": mesh_laplacian_smoothing(meshes), "normal": mesh_normal_consistency(meshes), "arap": arap_loss(meshes, ref_meshes),
            "photo": l1(y_true * y_col.unsqueeze(-1), y_pred * y_col.unsqueeze(-1)),
            "albedo": albedo_reg(params["texture"], uv_mask=params["uv_mask"], std=1.0, dist=d_a),
            "normal_reg": normal_reg(params["normal_map"], uv_mask=params["uv_mask"], dist=d_n)}
    total = sum(v * H.LOSS_WEIGHTS[k] for k, v in loss.items())
    total.backward()
    # ---- oracle
    with torch.no_grad():
        _, rv = H.prepare_mesh(P, torch.tensor([0]), sc["model"], sc["topo"])
    oloss, ototal, aux = H.step_losses(P, fid, sc["model"], sc["topo"], tg, S, focal, rv, d_a.long(), d_n.long())
    ototal.backward()
    for k, v in oloss.items():
        assert abs(loss[k].item() - v.item()) <= 2e-5 * abs(v.item()) + 1e-8, (k, loss[k].item(), v.item())
    assert ((y_pred.detach().cpu() - aux["y_pred"]).abs().max(-1).values > 1e-4).float().mean() < 1e-3
    for k in ("pose", "cam", "verts_disps", "shape", "light_positions", "amb_ratio", "texture", "normal_map", "rot", "trans"):
        assert rel(params[k].grad.cpu(), P[k].grad) < 2e-3, (k, rel(params[k].grad.cpu(), P[k].grad))


def test_unshadowed_phong_renderer():
    """render_image with the phong renderer (self_shadow=False path, optimize_sequence.py:485-488)."""
    from harp_amd.manopth.manolayer import ManoLayer
    from harp_amd.optimize_sequence import get_mesh_subdivider, init_params
    from harp_amd.renderer import renderer_helper
    from harp_amd.structures import Meshes
    from harp_amd.utils.visualize import prepare_materials, prepare_mesh, render_image
    from oracle import harp_ref as H
    sc = make_scene(T=2, S=96, seed=2)
    S, focal = sc["S"], sc["focal"]
    configs = dict(model_type="harp")
    layer = ManoLayer(flat_hand_mean=False, use_pca=False, model=sc["model_np"], device=DEV)
    sub = get_mesh_subdivider(layer, device=DEV)
    params = init_params(sc["seq"], True, True, None, layer.th_faces, False, torch.from_numpy(sc["tpl"]["verts_uvs"])[None],
                         torch.from_numpy(sc["tpl"]["faces_uvs"])[None], configs=configs, device=DEV, uv_mask=sc["uv_mask"])
    fid = torch.tensor([0, 1])
    lp = params["light_positions"][0].repeat(2, 1)
    phong, _, _ = renderer_helper.get_renderers(image_size=S, light_posi=lp, device=DEV)
    j, v, f, t = prepare_mesh(params, fid, layer, False, sub, False, configs, device=DEV)
    img = render_image(Meshes(v, f, t), params["cam"][fid.to(DEV)], 2, phong, S, focal, silhouette=False, device=DEV,
                       materials_properties=prepare_materials(params, 2, device=DEV))
    P = oracle_params(sc, params)
    with torch.no_grad():
        _, ov = H.prepare_mesh(P, fid, sc["model"], sc["topo"])
        ref = H.render_rgb(ov, sc["topo"], P, P["cam"][fid], S, focal, self_shadow=False)
    assert ((img.detach().cpu() - ref).abs().max(-1).values > 1e-4).float().mean() < 1e-3


def test_fit_entry_point_and_checkpoint(tmp_path):
    """optimize_hand_sequence for a few epochs across all three stages: loss decreases in the coarse stage, results are saved
    in the reference's saved_params.pkl layout and load back."""
    from harp_amd.manopth.manolayer import ManoLayer
    from harp_amd.optimize_sequence import optimize_hand_sequence
    from harp_amd.utils import file_utils
    from harp_amd.utils.config_utils import get_config
    sc = make_scene(T=5, S=96, seed=3)
    S = sc["S"]
    cfg = get_config(write_yaml=True, use_arm=False, img_size=S, focal_length=sc["focal"], total_epoch=6, training_stage=[3, 2, 1],
                     base_output_dir=str(tmp_path) + "/")
    layer = ManoLayer(flat_hand_mean=False, use_pca=False, model=sc["model_np"], device=DEV)
    tg = sc["targets"]
    ds = [(i, tg["y_true"][i], tg["y_sil"][i][..., None], tg["y_sil_col"][i][..., None]) for i in range(5)]
    hist = []
    params = optimize_hand_sequence(cfg, sc["seq"], ds, None, None, layer, torch.from_numpy(sc["tpl"]["verts_uvs"])[None],
                                    torch.from_numpy(sc["tpl"]["faces_uvs"])[None], device=DEV, uv_mask=sc["uv_mask"], batch_size=2,
                                    log_fn=lambda e, l, eng: hist.append(l))
    assert len(hist) == 6 and all(np.isfinite(hist)) and hist[2] < hist[0]
    assert not torch.equal(params["pose"].cpu(), sc["seq"]["pose"]) and not torch.equal(params["texture"].cpu(), torch.full_like(params["texture"].cpu(), 0))
    loaded = file_utils.load_result(str(tmp_path), device=DEV)
    for k in ("pose", "cam", "texture", "normal_map", "verts_disps", "shape"):
        assert torch.equal(loaded[k].detach().cpu(), params[k].cpu()), k
    assert isinstance(loaded["texture"], torch.nn.Parameter) and loaded["verts_disps"].is_cuda


def test_resume_and_known_appearance(tmp_path):
    """start_from / known_appearance (optimize_sequence.py:355-389, 264-289): the checkpoint is restored with the reference's
    re-initialisation (trans / rot collapsed to their means), the appearance and shape parameters stay frozen, pose / camera / light
    keep moving, and the result is saved as saved_params_test.pkl."""
    from harp_amd.manopth.manolayer import ManoLayer
    from harp_amd.optimize_sequence import optimize_hand_sequence
    from harp_amd.utils import file_utils
    from harp_amd.utils.config_utils import get_config
    sc = make_scene(T=5, S=96, seed=4)
    S = sc["S"]
    layer = ManoLayer(flat_hand_mean=False, use_pca=False, model=sc["model_np"], device=DEV)
    tg = sc["targets"]
    ds = [(i, tg["y_true"][i], tg["y_sil"][i][..., None], tg["y_sil_col"][i]) for i in range(5)]
    uvs = (torch.from_numpy(sc["tpl"]["verts_uvs"])[None], torch.from_numpy(sc["tpl"]["faces_uvs"])[None])
    first = str(tmp_path / "first") + "/"
    import os
    os.makedirs(first)
    cfg = get_config(write_yaml=False, use_arm=False, img_size=S, focal_length=sc["focal"], total_epoch=3, training_stage=[1, 1, 1], base_output_dir=first)
    p1 = optimize_hand_sequence(cfg, sc["seq"], ds, None, None, layer, *uvs, device=DEV, uv_mask=sc["uv_mask"], batch_size=2)
    second = str(tmp_path / "second") + "/"
    os.makedirs(second)
    cfg2 = get_config(write_yaml=False, use_arm=False, img_size=S, focal_length=sc["focal"], total_epoch=2, training_stage=[0, 2, 0],
                      base_output_dir=second, start_from=first, known_appearance=True, pose_already_opt=True)
    # pose_already_opt reads saved_params_test.pkl: make the first run's result available under that name too
    file_utils.save_result(p1, first, test=True)
    p2 = optimize_hand_sequence(cfg2, sc["seq"], ds, None, None, layer, *uvs, device=DEV, uv_mask=sc["uv_mask"], batch_size=2)
    for k in ("texture", "normal_map", "verts_disps", "shape"):
        assert torch.equal(p2[k].cpu(), p1[k].cpu()), k                       # frozen (bit-exact: zero gradient -> zero Adam update)
    assert not torch.equal(p2["pose"].cpu(), p1["pose"].cpu()) and not torch.equal(p2["light_positions"].cpu(), p1["light_positions"].cpu())
    assert torch.equal(p2["trans"].cpu(), (torch.zeros_like(p1["trans"]) + p1["trans"].mean(0)).cpu())      # no optimiser on trans
    assert os.path.exists(second + "saved_params_test.pkl")


def test_smooth_losses_through_hip_layer(golden_dir):
    """harp_amd.loss.smooth (one batched HIP LBS call for the 3-frame window) vs the golden values / gradients produced by the
    reference's loss/smooth.py + ManoLayer (tests/golden/smooth.npz)."""
    import os
    from harp_amd import synth
    from harp_amd.loss.smooth import LossSmoothPoses, LossSmoothRoots
    from harp_amd.manopth.manolayer import ManoLayer
    d = np.load(os.path.join(golden_dir, "smooth.npz"))
    tpl = synth.load_template("hand")
    layer = ManoLayer(flat_hand_mean=False, use_pca=False, model=synth.make_mano_model(tpl, seed=0), device=DEV)
    P = {k: torch.from_numpy(d[k]).to(DEV).requires_grad_(True) for k in ("rot", "pose", "shape", "trans", "cam")}
    fid = torch.from_numpy(d["fid"]).to(DEV)
    nF = int(d["n_frames"])
    lp = LossSmoothPoses(nF).smooth_pose(P, fid, layer, device=DEV)
    lr = LossSmoothRoots(nF, float(d["focal"]), int(d["res"])).smooth_root(P, fid, layer, device=DEV)
    assert abs(lp.item() - float(d["smooth_pose"])) <= 2e-5 * abs(float(d["smooth_pose"]))
    assert abs(lr.item() - float(d["smooth_root"])) <= 2e-5 * abs(float(d["smooth_root"]))
    (lp + 1e4 * lr).backward()
    for k in ("rot", "pose", "shape", "cam"):
        assert rel(P[k].grad.cpu(), torch.from_numpy(d["g_" + k])) < 5e-4, k
    # the root alignment cancels the translation exactly: d/d(trans) is pure rounding noise (~1e-3 against pose gradients of ~1e3)
    assert (P["trans"].grad.cpu() - torch.from_numpy(d["g_trans"])).abs().max() < 0.05


def test_fit_with_perceptual_term(tmp_path):
    """the entry point with the VGG term on (random filters from a torchvision-layout file named in the config): the appearance stages run
    eagerly, the reported epoch loss includes the term, the fit stays finite"""
    from harp_amd.manopth.manolayer import ManoLayer
    from harp_amd.model.vgg import Vgg16Features
    from harp_amd.optimize_sequence import optimize_hand_sequence
    from harp_amd.utils.config_utils import get_config
    sc = make_scene(T=4, S=96, seed=4)
    S = sc["S"]
    src = Vgg16Features(weights="random", seed=7)
    path = str(tmp_path / "vgg16.pth")
    torch.save({f"features.{k.split('.')[1]}.{k.split('.')[2]}": v for k, v in src.state_dict().items()}, path)
    layer = ManoLayer(flat_hand_mean=False, use_pca=False, model=sc["model_np"], device=DEV)
    tg = sc["targets"]
    ds = [(i, tg["y_true"][i], tg["y_sil"][i][..., None], tg["y_sil_col"][i][..., None]) for i in range(4)]
    hist = {}
    for use_vgg in (False, True):
        cfg = get_config(write_yaml=True, use_arm=False, img_size=S, focal_length=sc["focal"], total_epoch=3, training_stage=[0, 0, 3],
                         base_output_dir=str(tmp_path) + "/")
        if use_vgg:
            cfg["vgg_weights"] = path
        seen = []
        optimize_hand_sequence(cfg, sc["seq"], ds, None, None, layer, torch.from_numpy(sc["tpl"]["verts_uvs"])[None],
                               torch.from_numpy(sc["tpl"]["faces_uvs"])[None], device=DEV, uv_mask=sc["uv_mask"], batch_size=2,
                               log_fn=lambda e, l, eng: seen.append((l, eng.losses().get("vgg"))))
        hist[use_vgg] = seen
    assert all(v is None for _, v in hist[False]) and all(v is not None and v > 0 for _, v in hist[True])
    assert all(np.isfinite(l) for l, _ in hist[True])
