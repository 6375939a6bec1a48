"""Oracle (oracle/harp_ref.py) pinned against vectors produced by the reference itself
(tests/golden/make_golden.py imported manopth / loss.* from /root/reference in the build container)."""
import os

import numpy as np
import torch

from harp_amd import synth
from oracle import harp_ref as H


def _t(a):
    return torch.from_numpy(np.asarray(a))


def test_mano_forward_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "mano_layer.npz"))
    model = {k: _t(v) for k, v in synth.make_mano_model(seed=0).items()}
    pose, betas, trans = _t(g["pose"]).requires_grad_(), _t(g["betas"]).requires_grad_(), _t(g["trans"]).requires_grad_()
    verts, joints = H.mano_forward(model, pose, betas, trans)
    assert torch.allclose(verts, _t(g["verts"]), atol=2e-3, rtol=1e-5)      # mm
    assert torch.allclose(joints, _t(g["joints"]), atol=2e-3, rtol=1e-5)
    ((verts * _t(g["wv"])).sum() + (joints * _t(g["wj"])).sum()).backward()
    for name, p in (("g_pose", pose), ("g_betas", betas), ("g_trans", trans)):
        ref = _t(g[name])
        assert (p.grad - ref).norm() <= 1e-4 * ref.norm() + 1e-4, name
    v0, j0 = H.mano_forward(model, pose.detach(), betas.detach(), torch.zeros(pose.shape[0], 3))
    assert torch.allclose(v0, _t(g["verts_notrans"]), atol=2e-3)
    assert torch.allclose(j0, _t(g["joints_notrans"]), atol=2e-3)


def test_losses_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "losses.npz"))
    pr = _t(g["kps_pred"]).requires_grad_()
    lk = H.kps_loss(_t(g["kps_gt"]), pr)
    lk.backward()
    assert abs(lk.item() - float(g["kps_loss"])) <= 1e-6 * abs(float(g["kps_loss"]))
    assert torch.allclose(pr.grad, _t(g["kps_grad"]), rtol=1e-5, atol=1e-8)
    assert abs(H.kps_loss(_t(g["kps_gt"]), _t(g["kps_pred22"]), use_arm=True).item() - float(g["kps_loss_arm"])) < 1e-6

    topo = synth.build_topology(synth.load_template("hand")["faces0"], 778)
    cur = _t(g["arap_cur"]).requires_grad_()
    la = H.arap_loss(cur, _t(g["arap_ref"]), _t(topo["edges"]))
    la.backward()
    assert abs(la.item() - float(g["arap_loss"])) <= 1e-5 * abs(float(g["arap_loss"]))
    assert (cur.grad - _t(g["arap_grad"])).norm() <= 1e-5 * _t(g["arap_grad"]).norm()

    tex = _t(g["tex"]).requires_grad_()
    lt = H.albedo_reg(tex, _t(g["albedo_dist"]).long(), _t(g["mask"]))
    lt.backward()
    assert abs(lt.item() - float(g["albedo_loss"])) <= 1e-6
    assert torch.allclose(tex.grad, _t(g["albedo_grad"]), atol=1e-9)
    nm = _t(g["nm"]).requires_grad_()
    ln = H.normal_reg(nm, _t(g["normal_dist"]).long(), _t(g["mask"]))
    ln.backward()
    assert abs(ln.item() - float(g["normal_loss"])) <= 1e-6
    assert torch.allclose(nm.grad, _t(g["normal_grad"]), atol=1e-9)


def test_template_subdivision_kat():
    """SURVEY.md §4 item 1 was verified when the assets were built (tools/make_assets.py asserts it);
    here: sizes and that every subdivided face references one original corner + two midpoints (or 3 midpoints)."""
    for name, v0, e0, v, f, vt in (("hand", 778, 2315, 3093, 6152, 3327), ("arm", 1026, 3057, 4083, 8128, 4381)):
        tpl = synth.load_template(name)
        topo = synth.build_topology(tpl["faces0"], v0)
        assert topo["edges0"].shape == (e0, 2) and topo["n_verts"] == v and topo["faces"].shape == (f, 3)
        assert tpl["verts_uvs"].shape == (vt, 2) and tpl["faces_uvs"].shape == (f, 3)
        F0 = f // 4
        assert (topo["faces"][:3 * F0, 0] < v0).all() and (topo["faces"][:3 * F0, 1:] >= v0).all()
        assert (topo["faces"][3 * F0:] >= v0).all()


def test_smooth_losses_golden(golden_dir):
    """oracle restatement of loss/smooth.py vs values + gradients produced by the reference itself"""
    from harp_amd import synth
    from oracle import harp_ref as H
    d = np.load(os.path.join(golden_dir, "smooth.npz"))
    tpl = synth.load_template("hand")
    model = {k: torch.from_numpy(v) for k, v in synth.make_mano_model(tpl, seed=0).items()}
    P = {k: torch.from_numpy(d[k]).clone().requires_grad_(True) for k in ("rot", "pose", "shape", "trans", "cam")}
    fid = torch.from_numpy(d["fid"])
    nF = int(d["n_frames"])
    lp = H.smooth_pose_loss(P, fid, model, nF)
    lr = H.smooth_root_loss(P, fid, model, nF, float(d["focal"]), float(d["res"]))
    assert abs(lp.item() - float(d["smooth_pose"])) <= 1e-5 * abs(float(d["smooth_pose"]))
    assert abs(lr.item() - float(d["smooth_root"])) <= 1e-5 * abs(float(d["smooth_root"]))
    (lp + 1e4 * lr).backward()
    for k in ("rot", "pose", "shape", "trans", "cam"):
        g, want = P[k].grad, torch.from_numpy(d["g_" + k])
        assert (g - want).norm() <= 2e-4 * want.norm() + 1e-6, k
