"""MANO-to-METRO preprocessing fits (second half): mirror of the numeric part of the reference's
`metro_modifications/hand_utils.py` — `optimize_for_mano_param` (:16-131), `optimize_for_mano_arm_param` (:134-240), the loss
classes (:442-535), `optimize_smooth_seq` (:540-688) and the pickle helpers (`merge_dict`, `load_params`, `write_pkl`,
`remove_spike`).  Rendering / gif / mesh-export helpers are visualisation and stay out.

These loops are the other consumers of the LBS kernel: 1200 Adam iterations of (hand layer, MSE to the METRO vertices, backward) per
batch of frames.  In the reference every iteration is ~150 small torch launches; here `optimize_for_mano_param` is a native loop over
the C ABI — `harp_lbs_mano_fwd`, `harp_mse`, `harp_lbs_mano_bwd`, `harp_adam_tick/apply` on one flat parameter arena — captured
once per stage into a hipGraph of 7 nodes and replayed 500 + 700 times, for any number of frames at once.
"""
import ctypes
import os
import pickle

import numpy as np
import torch

from .. import _lib
from ..utils.data_util import combine_dict_to_batch

EPOCH_COARSE, EPOCH_FINE = 500, 700                      # hand_utils.py:22-23
_HYPER = np.dtype([("lr", "f4"), ("beta1", "f4"), ("beta2", "f4"), ("eps", "f4"), ("grad_scale", "f4"), ("step", "i4"),
                   ("step_size", "f4"), ("inv_sqrt_bc2", "f4")])               # harp_adam_hyper (include/harp_hip.h)


class ManoVertexFit:
    """Adam fit of (rot, pose, shape, trans) of B frames to target vertices (B,778,3) in mm, all frames in one launch sequence.
    Arena layout: [pose48 (B,48) = rot|pose | betas (B,10) | trans (B,3)]; same layout for gradient and both Adam moments."""

    def __init__(self, device_model, target_mm, trans_init=None):
        self.dm = device_model
        dev = target_mm.device
        B = self.B = target_mm.shape[0]
        self.target = target_mm.contiguous().float()
        self.n = B * 61
        self.p, self.g, self.m, self.v = (torch.zeros(self.n, dtype=torch.float32, device=dev) for _ in range(4))
        cut = lambda buf: (buf[:B * 48].view(B, 48), buf[B * 48:B * 58].view(B, 10), buf[B * 58:].view(B, 3))
        self.pose48, self.betas, self.trans = cut(self.p)
        self.g_pose48, self.g_betas, self.g_trans = cut(self.g)
        if trans_init is not None:
            self.trans.copy_(trans_init)
        L = _lib.lib()
        self.ws = torch.empty(L.harp_lbs_mano_ws_floats(B), dtype=torch.float32, device=dev)
        self.verts = torch.empty(B, 778, 3, dtype=torch.float32, device=dev)
        self.joints = torch.empty(B, 21, 3, dtype=torch.float32, device=dev)
        self.g_verts = torch.empty_like(self.verts)
        self.g_joints = torch.zeros_like(self.joints)
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self.hyper = torch.zeros(_HYPER.itemsize, dtype=torch.uint8, device=dev)

    def forward(self):
        p = _lib.ptr
        _lib.check(_lib.lib().harp_lbs_mano_fwd(ctypes.byref(self.dm.struct), p(self.pose48), p(self.betas), p(self.trans), self.B, p(self.ws),
                                                p(self.verts), p(self.joints), _lib.stream()), "harp_lbs_mano_fwd")

    def _iteration(self, global_only):
        L, p, st = _lib.lib(), _lib.ptr, _lib.stream
        self.forward()
        self.loss.zero_()
        _lib.check(L.harp_mse(p(self.verts), p(self.target), self.B * 778 * 3, p(self.loss), p(self.g_verts), st()), "harp_mse")
        _lib.check(L.harp_lbs_mano_bwd(ctypes.byref(self.dm.struct), p(self.pose48), p(self.betas), p(self.trans), self.B, p(self.ws),
                                       p(self.g_verts), p(self.g_joints), p(self.g_pose48), p(self.g_betas), p(self.g_trans), st()), "harp_lbs_mano_bwd")
        if global_only:
            # Adam([rot, trans]) (:74): the other parameters keep a zero gradient; with m = v = 0 their dense update is exactly 0
            self.g_pose48[:, 3:].zero_()
            self.g_betas.zero_()
        _lib.check(L.harp_adam_tick(self.hyper.data_ptr(), 1, st()), "harp_adam_tick")
        _lib.check(L.harp_adam_apply(p(self.p), p(self.g), p(self.m), p(self.v), self.n, self.hyper.data_ptr(), st()), "harp_adam_apply")

    def run(self, iterations, lr, global_only, use_graph=True):
        """a fresh torch.optim.Adam(lr) (:74, :96) stepped `iterations` times; returns the loss of the last iteration as a 0-d tensor"""
        h = np.zeros(1, _HYPER)
        h["lr"], h["beta1"], h["beta2"], h["eps"], h["grad_scale"] = lr, 0.9, 0.999, 1e-8, 1.0
        self.hyper.copy_(torch.from_numpy(h.view(np.uint8)))
        self.m.zero_(); self.v.zero_()
        if not use_graph or iterations < 3:
            for _ in range(iterations):
                self._iteration(global_only)
            return self.loss[0].clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self._iteration(global_only)                  # warm-up = iteration 1
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self._iteration(global_only)
        for _ in range(iterations - 1):
            graph.replay()
        return self.loss[0].clone()


def optimize_for_mano_param(pred_vertices, mano_layer, idx=0, vert_rgb=None, use_graph=True):
    """hand_utils.py:16-131.  pred_vertices (B,778,3) in metres -> dict of numpy arrays {joints, verts (mm), rot, pose, shape, trans}.
    The reference restarts up to four times while the final error exceeds 10.0 — from the same all-zero initialisation, i.e. every
    restart reproduces the first run; one run is done here and the verdict is printed the same way."""
    device = torch.device("cuda")
    target = pred_vertices.detach().to(device).float() * 1000.0
    mano_layer = mano_layer.to(device)
    fit = ManoVertexFit(mano_layer.device_model, target, trans_init=target.mean(1) / 1000.0)          # :62
    loss = fit.run(EPOCH_COARSE, 1e-1, global_only=True, use_graph=use_graph)                         # :74-87
    print("After coarse alignment: %6f" % float(loss))
    loss = fit.run(EPOCH_FINE, 1e-2, global_only=False, use_graph=use_graph)                          # :96-104
    print("After fine alignment: %6f" % float(loss))
    if float(loss) > 10.0:
        print("<<<<<<<<<<<<<<<<<<< ERROR TOO HIGH >>>>>>>>>>>>>>>>>>>>>>>")
    fit.forward()                                                                                     # :114
    out = lambda t: t.detach().cpu().numpy().copy()
    return {"joints": out(fit.joints), "verts": out(fit.verts), "rot": out(fit.pose48[:, :3]), "pose": out(fit.pose48[:, 3:]),
            "shape": out(fit.betas), "trans": out(fit.trans)}


def optimize_for_mano_arm_param(pred_vertices, mano_arm_layer, idx=0, vert_rgb=None):
    """hand_utils.py:134-240: the same two-stage fit through the SMPL-X arm layer's `return_type='mano'` vertices (HIP tree LBS behind
    torch autograd + torch.optim.Adam; translation starts at zero, :173)."""
    device = torch.device("cuda")
    target = pred_vertices.detach().to(device).float() * 1000.0
    B = target.shape[0]
    shape, rot, pose, trans = (torch.zeros(B, n, device=device, requires_grad=True) for n in (10, 3, 45, 3))
    layer = lambda: mano_arm_layer(betas=shape, global_orient=rot, transl=trans, right_hand_pose=pose, return_type="mano")
    mse = torch.nn.MSELoss()
    for group, lr, iters, tag in (([rot, trans], 1e-1, EPOCH_COARSE, "coarse"), ([rot, pose, shape, trans], 1e-2, EPOCH_FINE, "fine")):
        opt = torch.optim.Adam(group, lr=lr)
        for _ in range(iters):
            loss = mse(layer()[0], target)
            opt.zero_grad()
            loss.backward()
            opt.step()
        print("After %s alignment: %6f" % (tag, loss.item()))
    if loss.item() > 10.0:
        print("<<<<<<<<<<<<<<<<<<< ERROR TOO HIGH >>>>>>>>>>>>>>>>>>>>>>>")
    with torch.no_grad():
        verts, joints = layer()
    out = lambda t: t.detach().cpu().numpy().copy()
    return {"joints": out(joints), "verts": out(verts), "rot": out(rot), "pose": out(pose), "shape": out(shape), "trans": out(trans)}


# ---- pickle helpers --------------------------------------------------------------------------------------------------
merge_dict = combine_dict_to_batch                       # hand_utils.py:363-381 is utils/data_util.py:54-73 verbatim


def load_params(mano_dir):
    """hand_utils.py:384-398: every *.pkl of the directory, sorted by name, stacked"""
    frames = []
    for name in sorted(n for n in os.listdir(mano_dir) if n.endswith(".pkl")):
        with open(os.path.join(mano_dir, name), "rb") as f:
            frames.append(pickle.load(f))
    return merge_dict(frames)


def write_pkl(params, pkl_out_dir, unscreen=True):
    """hand_utils.py:766-782: one `%04d_mano.pkl` per frame (numbered from 1 with unscreen), every entry with a leading singleton axis
    except 'cam'"""
    first = 1 if unscreen else 0
    for i in range(len(params["cam"])):
        rec = {k: (v[i] if k == "cam" else v[i, None]).detach().cpu().numpy() for k, v in params.items()}
        with open(os.path.join(pkl_out_dir, "%04d_mano.pkl" % (i + first)), "wb") as f:
            pickle.dump(rec, f)


def remove_spike(params):
    """hand_utils.py:785-801: a frame whose pose jumps by more than 1.0 (L2) from its predecessor AND to its successor is replaced by
    the mean of the two"""
    pose = params["pose"]
    jump = torch.norm(pose[1:] - pose[:-1], dim=1) > 1.0
    spike = torch.zeros(len(pose), dtype=torch.bool)
    spike[1:-1] = jump[:-1] & jump[1:]
    fixed = pose.clone().detach()
    fixed[1:-1] = torch.where(spike[1:-1, None], (pose[:-2] + pose[2:]) / 2.0, pose[1:-1])
    params["pose"] = fixed
    return params


# ---- loss terms of the sequence smoothing (hand_utils.py:442-535) ---------------------------------------------------------
def func_l2(x):
    return torch.sum(x ** 2)


class LossInit:
    def __init__(self, params, device="cuda"):
        self.poses, self.shapes = torch.Tensor(params["poses"]).to(device), torch.Tensor(params["shapes"]).to(device)

    def init_poses(self, poses, **kwargs):
        return func_l2(poses - self.poses) / poses.shape[0]

    def init_shapes(self, shapes, **kwargs):
        return func_l2(shapes - self.shapes) / shapes.shape[0]


class LossAnchor:
    """sum of squared distances to fixed key points / number of frames (:481-496)"""

    def __init__(self, keypoints3d, device="cuda", norm="l2"):
        self.keypoints3d = torch.Tensor(keypoints3d).to(device)[..., :3]
        self.nFrames, self.nJoints, self.norm = self.keypoints3d.shape[0], self.keypoints3d.shape[1], norm

    def loss_func(self, kpts_est, **kwargs):
        nj = min(kpts_est.shape[1], self.keypoints3d.shape[1], 21)
        return func_l2(kpts_est[:, :nj, :3] - self.keypoints3d[:, :nj, :3]) / self.nFrames


class LossKeypoints3D(LossAnchor):
    """the same against root-relative key points (:461-478)"""

    def __init__(self, keypoints3d, device="cuda", norm="l2"):
        k = torch.Tensor(keypoints3d)
        super().__init__(k - k[:, 0, :].unsqueeze(1), device=device, norm=norm)


class LossSmoothPoses:
    """distance to the detached mean of the 3-frame window, per view (:499-513)"""

    def __init__(self, nViews, nFrames):
        self.nViews, self.nFrames, self.norm = nViews, nFrames, "l2"

    def poses(self, poses, **kwargs):
        total = 0
        for view in poses[:self.nViews * self.nFrames].split(self.nFrames):
            d = view.detach()
            total = total + func_l2(view[1:-1] - (d[1:-1] + d[:-2] + d[2:]) / 3)
        return total / (self.nFrames - 2) / self.nViews


class LossSmoothBodyMean:
    """distance to the detached midpoint of the two neighbours (:516-524)"""

    def body(self, kpts_est, **kwargs):
        d = kpts_est.detach()
        return func_l2(kpts_est[1:-1] - (d[:-2] + d[2:]) / 2) / (kpts_est.shape[0] - 2)


LossSmoothCam = LossSmoothBodyMean                      # :527-535 is the same term


def optimize_smooth_seq(params_cpu, mano_layer, use_smplx_arm=False, nimble_hand=False, img_res=224, mano_j_regressor=None,
                        total_iter_pose=1000, total_iter_cam=1000):
    """hand_utils.py:540-688.  Stage 1: Adam(lr 1e-3) on rot / pose / shape against 1e-2 * root-relative key-point anchor + 1e-1 *
    3-frame smoothness of the root-relative joints, stopped when the running-average loss stops falling by 1e-5.  Stage 2: Adam(lr
    1e-3, ReduceLROnPlateau(patience 10)) on the camera only, anchor + smoothness of camera translation + root.  Like the reference,
    stage 2 adds the (N,1,3) camera translation to the (N,3) roots, which broadcasts to (N,N,3); that is kept, since it sets the
    relative weight of the two terms.  Returns CPU tensors incl. refreshed 'joints' / 'verts'."""
    if nimble_hand:
        raise NotImplementedError("model_type 'nimble' is out of scope (SURVEY.md §8)")
    device = torch.device("cuda")
    mano_layer = mano_layer.to(device) if not use_smplx_arm else mano_layer
    params_cpu["cam"] = params_cpu["cam"].unsqueeze(1)
    params = {k: torch.Tensor(v).to(device) for k, v in params_cpu.items()}
    learn = [params["rot"], params["pose"], params["shape"], params["cam"]]
    for t in learn:
        t.requires_grad = True
    n_frames = len(params["pose"])

    def layer():
        if use_smplx_arm:
            return mano_layer(betas=params["shape"], global_orient=params["rot"], transl=params["trans"], right_hand_pose=params["pose"],
                              return_type="mano")
        return mano_layer(torch.cat((params["rot"], params["pose"]), 1), params["shape"], params["trans"])

    anchor, smooth = LossKeypoints3D(params_cpu["joints"], device=device), LossSmoothPoses(1, n_frames)
    opt = torch.optim.Adam(learn, lr=1e-3)
    prev = 999999.0
    for it in range(total_iter_pose):
        joints = layer()[1]
        joints = joints - joints[:, 0, :].unsqueeze(1)
        loss = 1e-2 * anchor.loss_func(joints) + 1e-1 * smooth.poses(joints)          # 'smooth_body' has weight 0 (:570)
        if it > 0 and prev - loss.item() < 0.00001:
            break
        prev = (prev + loss.item()) / 2.0
        opt.zero_grad()
        loss.backward()
        opt.step()
    with torch.no_grad():
        verts, joints = layer()
    params["joints"], params["verts"] = joints, verts
    focal = 1000.0 * (img_res / 224.0)
    root = joints[:, 0, :].detach() / 1000.0                                            # independent of the camera: evaluated once

    def cam_rel_root():
        cam = params["cam"]
        cam_t = torch.stack([cam[:, :, 1], cam[:, :, 2], 2 * focal / (img_res * cam[:, :, 0] + 1e-9)], dim=2)
        return cam_t + root                                                             # (N,1,3) + (N,3) -> (N,N,3), see above

    anchor, smooth = LossAnchor(cam_rel_root().detach(), device=device), LossSmoothPoses(1, n_frames)
    opt = torch.optim.Adam([params["cam"]], lr=1e-3)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, patience=10)
    for it in range(total_iter_cam):
        x = cam_rel_root()
        loss = 1e-2 * anchor.loss_func(x) + 1e-2 * smooth.poses(x)
        opt.zero_grad()
        loss.backward()
        opt.step()
        sched.step(loss)
    params["cam"] = params["cam"].squeeze(1)
    return {k: v.detach().cpu() for k, v in params.items()}
