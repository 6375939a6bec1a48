"""Shared synthetic scene for the parity tests (seeded; small enough for the CPU oracle to finish in seconds)."""
import numpy as np
import torch

from harp_amd import synth
from oracle import harp_ref as H


def make_scene(T=3, S=128, seed=0):
    torch.manual_seed(seed)
    tpl = synth.load_template("hand")
    topo_np = synth.build_topology(tpl["faces0"], 778)
    model_np = synth.make_mano_model(tpl, seed=seed)
    model = {k: torch.from_numpy(v) for k, v in model_np.items()}
    topo = {k: torch.from_numpy(np.asarray(v)).long() if isinstance(v, np.ndarray) else v for k, v in topo_np.items()}
    seq, focal = synth.make_sequence(model_np, T, S, seed=seed)
    with torch.no_grad():
        _, j = H.mano_forward(model, torch.cat((seq["rot"], seq["pose"]), 1), seq["shape"].mean(0).repeat(T, 1), seq["trans"])
    seq["joints"] = j + torch.randn_like(j) * 3.0
    uv_mask = torch.from_numpy(tpl["uv_mask"]).double() / 255
    targets = dict(y_true=torch.rand(T, S, S, 3), y_sil=(torch.rand(T, S, S) > 0.5).float(), y_sil_col=(torch.rand(T, S, S) > 0.4).float())
    return dict(tpl=tpl, topo_np=topo_np, model_np=model_np, model=model, topo=topo, seq=seq, focal=focal, uv_mask=uv_mask,
                targets=targets, T=T, S=S)


def oracle_params(sc, source, dtype=torch.float32):
    """dict of leaf tensors (requires_grad) cloned from `source` (engine.params or any dict of tensors)."""
    keys = ("pose", "cam", "verts_disps", "shape", "light_positions", "amb_ratio", "texture", "normal_map", "rot", "trans")
    P = {k: source[k].detach().cpu().to(dtype).clone().requires_grad_() for k in keys}
    P.update(verts_uvs=torch.from_numpy(sc["tpl"]["verts_uvs"]).to(dtype), faces_uvs=torch.from_numpy(sc["tpl"]["faces_uvs"]).long(),
             uv_mask=sc["uv_mask"], init_joints=sc["seq"]["joints"].to(dtype))
    return P


def scene_f64(sc, targets):
    """(model, targets) of a make_scene() scene in float64 for the oracle"""
    model = {k: (v.double() if v.is_floating_point() else v) for k, v in sc["model"].items()}
    return model, {k: v.double() for k, v in targets.items()}


# Fraction of the covered pixels the ambiguous-pixel mask removed, per test case, as MEASURED on MI355X (rounds 3 and 5; every caller prints
# its value).  A case fails when its fraction exceeds 1.5 x the recorded one: the mask may not quietly grow to hide kernel errors.
# Every masked comparison has an UNMASKED companion with a looser gradient bound (tests/test_gpu_baseline.py), so what the mask
# removes is bounded, not ignored.
AMBIGUOUS_FRACTION = {
    "api_loop_body_128": 0.0535,
    "app_only_hand_256_b2": 0.0618,
    "c2_hand_128_b18": 0.0436,
    "c2_hand_512_b18": 0.0466,
    "c2c3_hand_512_b2": 0.0602,
    "c3_hand_512_b32": 0.0515,
    "c5_arm_1024_b1": 0.0262,
    "c5_arm_1024_b32": 0.0311,
    "c5_arm_1024_b8": 0.0266,
    "parity_arm_128": 0.0245,
    "parity_empty_supertiles_256": 0.0421,
    "parity_full_step_128": 0.0841,
    "parity_stage_128_shadow0": 0.0094,
    "parity_stage_128_shadow1": 0.0572,
    "parity_vgg_128": 0.0659,
    "smoke_hand_256_b1": 0.0649,
    "ten_steps_hand_128": 0.0821,
}


def check_removed(tag, removed):
    import json, os
    print(f"[ambiguous-pixel mask] {tag}: {removed:.4f} of the covered pixels removed")
    log = os.environ.get("HARP_MASK_LOG")
    if log:
        with open(log, "a") as f:
            f.write(json.dumps({"tag": tag, "removed": removed}) + "\n")
    if os.environ.get("HARP_MASK_RECORD") == "1":       # measuring run (fills AMBIGUOUS_FRACTION): log only
        return removed
    rec = AMBIGUOUS_FRACTION.get(tag)
    assert rec is not None, f"no recorded mask fraction for {tag!r} (measured now: {removed:.4f})"
    assert removed <= 1.5 * rec + 1e-4, (tag, removed, rec)
    return removed


def rel(a, b):
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def ambiguous_pixels(P, model, topo, S, focal, fid, y_true=None, use_arm=False, self_shadow=True):
    """(len(fid),S,S) bool: covered pixels whose colour is NOT decided at float32 precision (see mask_ambiguous_pixels), flagged by the
    oracle's float64 forward pass for the parameters P (any dtype; evaluated in float64)."""
    f64 = lambda d: {k: (v.detach().double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in d.items()}
    P64, model64 = f64(P), f64(model)
    fid = torch.as_tensor(fid)
    with torch.no_grad():
        _, verts = H.prepare_mesh(P64, fid, model64, topo, use_arm=use_arm)
        img, aux = H.render_rgb(verts, topo, P64, P64["cam"][fid], S, focal, self_shadow=self_shadow, return_aux=True, flag_ambiguous=True)
    amb = aux["ambiguous"]
    if y_true is not None:      # kink of the L1 photometric term: |y_pred - y_true| below what float32 resolves flips the sign of a channel's gradient
        amb = amb | ((img - y_true[fid].double()).abs() < 3e-4).any(-1)
    return amb & (aux["pix_to_face"][..., 0] >= 0), aux


def mask_scene_targets(sc, params, fid):
    """make_scene targets with the ambiguous pixels of frames `fid` (under `params`) taken out of the photometric mask: returns a new
    targets dict and the fraction of covered pixels removed"""
    P = oracle_params(sc, params)
    amb, aux = ambiguous_pixels(P, sc["model"], sc["topo"], sc["S"], sc["focal"], fid, sc["targets"]["y_true"])
    tg = dict(sc["targets"])
    col = tg["y_sil_col"].clone()
    for i, f in enumerate(torch.as_tensor(fid).tolist()):
        col[f][amb[i]] = 0.0
    tg["y_sil_col"] = col
    return tg, amb.sum().item() / max((aux["pix_to_face"][..., 0] >= 0).sum().item(), 1)


# ----------------------------------------------------------------------------------------------------------------------
# Fit cases at the BASELINE.json sizes: an engine on the GPU + everything the fp64 oracle needs for the same inputs.
# ----------------------------------------------------------------------------------------------------------------------
def erode(mask, iters=2):
    """3x3 erosion x2 (utils/data_util.py:17-20)"""
    m = mask[:, None]
    for _ in range(iters):
        m = -torch.nn.functional.max_pool2d(-m, 3, stride=1, padding=1)
    return m[:, 0]


def make_fit_case(kind="hand", T=2, S=512, B=2, seed=0, device="cuda", raw=False, **engine_kw):
    """FitEngine for `kind` in {"hand", "arm"} (raw=True: the UN-subdivided 778-vertex MANO mesh, config C1) with non-trivial
    parameters and REALISTIC targets — rendered by the engine from a perturbed "ground-truth" parameter set like bench.py does
    (SURVEY.md §8d), so the silhouette / photometric gradients are coherent rather than noise.  Returns a dict with the engine
    and the CPU tensors the oracle consumes."""
    from harp_amd.engine import FitEngine
    g = torch.Generator().manual_seed(seed + 100)
    if kind == "hand":
        tpl = synth.load_template("hand")
        topo_np = synth.build_raw_topology(tpl["faces0"], 778) if raw else synth.build_topology(tpl["faces0"], 778)
        model_np = synth.make_mano_model(tpl, seed=seed)
        seq, focal = synth.make_sequence(model_np, T, S, seed=seed)
        kw = {}
    else:
        tpl = synth.load_template("arm")
        topo_np = synth.build_topology(tpl["faces0"], 1026)
        model_np = synth.make_smplx_arm_model(tpl, seed=seed)
        focal = 1000.0 * S / 224.0
        c = model_np["v_template"].mean(0)
        seq = dict(pose=torch.randn(T, 45, generator=g) * 0.15, rot=torch.randn(T, 3, generator=g) * 0.2, trans=torch.randn(T, 3, generator=g) * 0.01,
                   shape=torch.randn(T, 10, generator=g) * 0.3,
                   cam=torch.tensor([[2 * focal / (S * 1.6), -float(c[0]), -float(c[1])]]).repeat(T, 1) + torch.randn(T, 3, generator=g) * 0.005)
        kw = dict(use_arm=True, opt_arm_pose=True)
    kw.update(engine_kw)
    V = int(topo_np["n_verts"])
    if raw:      # no UV layout for the raw mesh (C1 is silhouette-only): dummy tables, never sampled by a coarse-only stage
        verts_uvs, faces_uvs = np.zeros((1, 2), np.float32), np.zeros((topo_np["faces"].shape[0], 3), np.int32)
    else:
        verts_uvs, faces_uvs = tpl["verts_uvs"], tpl["faces_uvs"]
    uv_mask = torch.from_numpy(tpl["uv_mask"]).double() / 255
    seq["joints"] = torch.zeros(T, 21, 3)
    eng = FitEngine(model_np, topo_np, verts_uvs, faces_uvs, uv_mask.float(), seq, S, focal, B, device=device, **kw)
    # ---- "ground truth": perturbed pose / camera / shape / displacement / texture -> rendered targets
    cur = {k: eng.params[k].clone() for k in ("pose", "cam", "shape", "verts_disps", "texture", "normal_map", "trans", "wrist_pose")}
    dev = eng.dev
    with torch.no_grad():
        eng.params["pose"].add_((torch.randn(T, 45, generator=g) * 0.05).to(dev))
        eng.params["cam"][:, 1:].add_((torch.randn(T, 2, generator=g) * 0.004).to(dev))
        eng.params["shape"].add_((torch.randn(10, generator=g) * 0.3).to(dev))
        eng.params["verts_disps"].copy_((torch.randn(V, 1, generator=g) * 0.0008).to(dev))
        tex = torch.nn.functional.interpolate(torch.rand(1, 3, 32, 32, generator=g), size=512, mode="bilinear")[0].permute(1, 2, 0)
        eng.params["texture"].copy_((0.35 + 0.5 * tex)[None].to(dev))
    y_true = torch.empty(T, S, S, 3, device=dev)
    y_sil = torch.empty(T, S, S, device=dev)
    joints = torch.empty(T, eng.n_joints, 3, device=dev)
    eng.y_true, eng.y_sil, eng.y_sil_col = y_true, y_sil, y_sil          # placeholders: the losses of this pass are ignored
    eng.set_stage(False, not raw)
    for s0 in range(0, T, B):
        n = min(B, T - s0)
        eng.fid[:n].copy_(torch.arange(s0, s0 + n, dtype=torch.int32).to(dev))
        eng.tfid.zero_()
        eng.forward_backward(coarse=True, app=not raw, B=n)
        joints[s0:s0 + n] = eng.s["joints_mm"][:n]
        y_sil[s0:s0 + n] = (eng.s["alpha"][:n] > 0.5).float()
        if not raw:
            y_true[s0:s0 + n] = eng.s["rgb"][:n]
    if raw:
        y_true.fill_(1.0)
    torch.cuda.synchronize()
    # ---- the parameters the step is evaluated at: the initial ones plus a displacement / texture / normal-map / translation state
    with torch.no_grad():
        for k, v in cur.items():
            eng.params[k].copy_(v)
        eng.params["verts_disps"].copy_((torch.randn(V, 1, generator=g) * 0.0006).to(dev))
        if not raw:
            eng.params["texture"].copy_((torch.rand(1, 512, 512, 3, generator=g) * 0.3 + 0.45).to(dev))
            eng.params["normal_map"].copy_((torch.tensor([0., 0., 1.]).repeat(1, 512, 512, 1) + torch.randn(1, 512, 512, 3, generator=g) * 0.1).to(dev))
        eng.params["trans"].copy_((torch.randn(T, 3, generator=g) * 0.01).to(dev))
        if kind == "arm":
            eng.params["wrist_pose"].copy_((torch.randn(T, 3, generator=g) * 0.2).to(dev))
    init_joints = (joints[:, :21] + torch.randn(T, 21, 3, generator=g).to(dev) * 2.0).contiguous()    # METRO-like noisy anchors (mm)
    eng.init_joints = init_joints                                # (T,21,3) also for the arm: kps_loss drops its 22nd joint (loss/kps_loss.py:7-8)
    y_col = erode(y_sil)
    eng.set_targets(y_true, y_sil, y_col)
    eng.compute_reference_mesh()
    eng.g_buf.zero_()
    targets = dict(y_true=y_true.cpu(), y_sil=y_sil.cpu(), y_sil_col=y_col.cpu())
    model = {k: torch.from_numpy(np.asarray(v)) for k, v in model_np.items()}
    topo = {k: torch.from_numpy(np.asarray(v)).long() if isinstance(v, np.ndarray) else v for k, v in topo_np.items()}
    return dict(eng=eng, kind=kind, model=model, topo=topo, tpl=tpl, targets=targets, focal=focal, S=S, T=T, B=B, uv_mask=uv_mask,
                init_joints=init_joints.cpu(), verts_uvs=torch.from_numpy(np.asarray(verts_uvs, np.float32)),
                faces_uvs=torch.from_numpy(np.asarray(faces_uvs)).long())


ORACLE_KEYS = ("pose", "cam", "verts_disps", "shape", "light_positions", "amb_ratio", "texture", "normal_map", "rot", "trans", "wrist_pose")


def oracle_inputs(case, dtype=torch.float64):
    """leaf tensors (requires_grad) cloned from the engine's parameters in `dtype`, plus model / targets in the same dtype"""
    eng = case["eng"]
    P = {k: eng.params[k].detach().cpu().to(dtype).clone().requires_grad_() for k in ORACLE_KEYS}
    P.update(verts_uvs=case["verts_uvs"].to(dtype), faces_uvs=case["faces_uvs"], uv_mask=case["uv_mask"], init_joints=case["init_joints"].to(dtype))
    model = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in case["model"].items()}
    targets = {k: v.to(dtype) for k, v in case["targets"].items()}
    return P, model, targets


def oracle_step(case, fid, dtype=torch.float64, coarse=True, app=True, P=None, model=None, targets=None, terms=None):
    """loss dict, weighted total (backward() already run -> P[k].grad), aux images of the oracle for the engine's current parameters and
    texture-regulariser offsets.  terms: restrict the objective to these loss names (e.g. ("silhouette",) for config C1)."""
    eng = case["eng"]
    if P is None:
        P, model, targets = oracle_inputs(case, dtype)
    use_arm = case["kind"] == "arm"
    rv = case.get(("ref_verts", dtype))
    if rv is None:          # ARAP reference = frame 0 under the INITIAL parameters (optimize_sequence.py:429-435): computed once, like the engine
        with torch.no_grad():
            _, rv = H.prepare_mesh(P, torch.tensor([0]), model, case["topo"], use_arm=use_arm)
        case[("ref_verts", dtype)] = rv
    loss, total, aux = H.step_losses(P, fid, model, case["topo"], targets, case["S"], case["focal"], rv, eng.dist_albedo.cpu().long(),
                                     eng.dist_normal.cpu().long(), coarse=coarse, app=app, self_shadow=eng.self_shadow, use_arm=use_arm)
    if terms is not None:
        total = sum(loss[k] * H.LOSS_WEIGHTS[k] for k in terms)
    total.backward()
    return P, loss, total, aux, rv


def engine_eval(case, fid, coarse=True, app=True, tfid=None):
    """one forward + backward of the engine on frames `fid` with the current offsets (no Adam)"""
    eng = case["eng"]
    n = len(fid)
    eng.fid[:n].copy_(torch.as_tensor(fid).int().to(eng.dev))
    eng.tfid[:n].copy_(torch.as_tensor(fid if tfid is None else tfid).int().to(eng.dev))
    eng.auto_draw = False
    eng.set_stage(coarse, app)
    eng.forward_backward(coarse, app, B=n)
    torch.cuda.synchronize()
    return eng.losses()


def mask_ambiguous_pixels(case):
    """Take pixels whose colour is NOT DECIDED AT FLOAT32 PRECISION out of the photometric mask of every frame (for the HIP path and the
    oracle alike).  HARP's K=1 passes are discontinuous where a pixel centre sits on a face edge — in the camera view, or in the light
    view, where an uncovered texel reads depth -1 = "in shadow" for up to 9 camera pixels — and where a hit point projects onto the
    .round() boundary of its shadow-tap index (renderer_helper.py:385): there any two float32 evaluations (ours, the oracle's in fp32, the
    reference's CUDA kernels) may differ by a whole tap (1/9 of the diffuse term).  The same holds for the other branch points of the
    path: depth ties between the two nearest faces, the sign branch of the tangent frame (pbr_materials.py:68), relu(n.l) at 0, the
    texel boundaries of the bilinear footprint (the derivative w.r.t. uv jumps) and the kink of the L1 term (|y_pred - y_true| ~ 0: the
    sign IS the gradient) — and for pixels on faces that are slivers in NDC, whose barycentric gradients (~1/area) amplify the float32
    rounding of the NDC vertices themselves (measured: one such face, seen edge-on at the silhouette, carried the largest entries of
    dL/d ndc with a 0.25 % error, 1e-2 after projection onto the pose parameters; position / normal / texture gradients of the same
    launch agreed to 2e-6).  One such pixel moves a gradient by ~1/sqrt(#pixels) = 0.2-1 % in relative L2 at these image sizes.
    The flags come from the oracle's float64 forward pass
    alone (oracle/p3d_like.rasterize_meshes(return_ambiguous=True), oracle/harp_ref.render_rgb(flag_ambiguous=True)); the oracle's own
    fp32-vs-fp64 image differences > 1e-3 all fall on flagged pixels.  Returns the fraction of covered pixels removed."""
    eng = case["eng"]
    P, model, targets = oracle_inputs(case, torch.float64)
    fid = torch.arange(case["T"])
    amb, aux = ambiguous_pixels(P, model, case["topo"], case["S"], case["focal"], fid, targets["y_true"], use_arm=case["kind"] == "arm",
                                self_shadow=eng.self_shadow)
    cov = (aux["pix_to_face"][..., 0] >= 0).sum().item()
    y_col = case["targets"]["y_sil_col"].clone()
    y_col[amb] = 0.0
    case["targets"]["y_sil_col"] = y_col
    eng.set_targets(case["targets"]["y_true"], case["targets"]["y_sil"], y_col)
    return amb.sum().item() / max(cov, 1)
