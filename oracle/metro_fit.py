"""TEST INFRASTRUCTURE ONLY — CPU oracle, part 3: the MANO-to-METRO preprocessing fits
(metro_modifications/hand_utils.py in /root/reference), restated with the oracle's MANO layer (oracle.harp_ref.mano_forward,
pinned against the reference ManoLayer by tests/golden/mano.npz) + torch autograd + torch.optim.Adam, exactly the reference's recipe.

PARITY UNPINNED for this file: the reference module imports cv2 / trimesh / imageio at module level (absent from the build image),
so it cannot be imported to produce golden vectors; the formulas below follow its source line by line instead.
Only tests/ may import this.
"""
import torch

from . import harp_ref as H


def fit_mano_to_vertices(model, pred_vertices_m, n_coarse=500, n_fine=700):
    """optimize_for_mano_param, hand_utils.py:16-131 (one try; all tries start from the same zeros)."""
    target = pred_vertices_m.detach().float() * 1000.0                                   # :39
    B = target.shape[0]
    shape = torch.zeros(B, 10, requires_grad=True)                                       # :55-62
    rot = torch.zeros(B, 3, requires_grad=True)
    pose = torch.zeros(B, 45, requires_grad=True)
    trans = (torch.zeros(B, 3) + target.mean(1) / 1000.0).requires_grad_()
    mse = torch.nn.MSELoss()
    losses = []
    for group, lr, n in (([rot, trans], 1e-1, n_coarse), ([rot, pose, shape, trans], 1e-2, n_fine)):     # :74, :96
        opt = torch.optim.Adam(group, lr=lr)
        for _ in range(n):
            verts, _ = H.mano_forward(model, torch.cat((rot, pose), 1), shape, trans)
            loss = mse(verts, target)
            opt.zero_grad()
            loss.backward()
            opt.step()
        losses.append(loss.item())
    with torch.no_grad():
        verts, joints = H.mano_forward(model, torch.cat((rot, pose), 1), shape, trans)   # :114
    return {"joints": joints, "verts": verts, "rot": rot.detach(), "pose": pose.detach(), "shape": shape.detach(),
            "trans": trans.detach()}, losses


def remove_spike(pose):
    """hand_utils.py:785-801"""
    step = torch.norm(pose[1:] - pose[:-1], dim=1)
    out = pose.clone()
    for i in range(1, len(pose) - 1):
        if step[i - 1] > 1.0 and step[i] > 1.0:
            out[i] = (pose[i - 1] + pose[i + 1]) / 2.0
    return out


def keypoint_term(est, anchor):
    """LossKeypoints3D / LossAnchor.loss_func, hand_utils.py:473-478, 491-496"""
    nj = min(est.shape[1], anchor.shape[1], 21)
    return torch.sum((est[:, :nj, :3] - anchor[:, :nj, :3]) ** 2) / anchor.shape[0]


def window_term(x, n_frames):
    """LossSmoothPoses(1, nFrames).poses, hand_utils.py:505-513"""
    x = x[:n_frames]
    interp = x.clone().detach()
    interp[1:-1] = (interp[1:-1] + interp[:-2] + interp[2:]) / 3
    return torch.sum((x[1:-1] - interp[1:-1]) ** 2) / (n_frames - 2)


def midpoint_term(x):
    """LossSmoothBodyMean.body, hand_utils.py:520-524"""
    interp = x.clone().detach()
    interp[1:-1] = (interp[:-2] + interp[2:]) / 2
    return torch.sum((x[1:-1] - interp[1:-1]) ** 2) / (x.shape[0] - 2)


def smooth_sequence(model, params_in, img_res=224, total_iter_pose=1000, total_iter_cam=1000):
    """optimize_smooth_seq for the MANO hand, hand_utils.py:540-688."""
    params = {k: torch.Tensor(v).clone() for k, v in params_in.items()}
    params["cam"] = params["cam"].unsqueeze(1)                                            # :548
    learn = [params["rot"], params["pose"], params["shape"], params["cam"]]
    for t in learn:
        t.requires_grad = True
    N = len(params["pose"])
    anchor = torch.Tensor(params_in["joints"])
    anchor = anchor - anchor[:, 0, :].unsqueeze(1)                                        # :464
    layer = lambda: H.mano_forward(model, torch.cat((params["rot"], params["pose"]), 1), params["shape"], params["trans"])
    opt = torch.optim.Adam(learn, lr=1e-3)                                                # :580
    prev = 999999.0
    for it in range(total_iter_pose):
        _, joints = layer()
        joints = joints - joints[:, 0, :].unsqueeze(1)
        loss = 1e-2 * keypoint_term(joints, anchor) + 0 * midpoint_term(joints) + 1e-1 * window_term(joints, N)      # :569-571
        if it > 0 and prev - loss.item() < 0.00001:                                       # :603
            break
        prev = (prev + loss.item()) / 2.0
        opt.zero_grad()
        loss.backward()
        opt.step()
    with torch.no_grad():
        verts, joints = layer()
    params["joints"], params["verts"] = joints, verts
    focal = 1000.0 * (img_res / 224.0)                                                    # :629-632

    def cam_rel_root():
        _, j = layer()
        cam = params["cam"]
        cam_t = torch.stack([cam[:, :, 1], cam[:, :, 2], 2 * focal / (img_res * cam[:, :, 0] + 1e-9)], dim=2)
        return cam_t + j[:, 0, :] / 1000.0                                                # :645-646 (broadcasts to (N,N,3))

    fixed = cam_rel_root().detach()
    opt = torch.optim.Adam([params["cam"]], lr=1e-3)                                      # :654
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, patience=10)
    for it in range(total_iter_cam):
        x = cam_rel_root()
        loss = 1e-2 * keypoint_term(x, fixed) + 1e-2 * window_term(x, N)                  # :649-651
        opt.zero_grad()
        loss.backward()
        opt.step()
        sched.step(loss)
    params["cam"] = params["cam"].squeeze(1)
    return {k: v.detach() for k, v in params.items()}
