"""MANO hand layer backed by the HIP LBS kernels (csrc/lbs.hip).

Mirror of the reference's `manopth.manolayer.ManoLayer` (manopth/manolayer.py:14-296) for the configuration HARP
instantiates — `ManoLayer(mano_root='mano/models', flat_hand_mean=False, use_pca=False)` (utils/hand_model_utils.py:74):
axis-angle root, axis-angle joints, right hand.  `forward(th_pose_coeffs, th_betas, th_trans)` keeps the reference's
argument meaning and returns (verts (B,778,3) mm, joints (B,21,3) mm).
"""
import ctypes
import os
import pickle

import numpy as np
import torch

from .. import _lib


class ManoDeviceModel:
    """Device copies of the MANO buffers in the layouts the kernels want (see harp_mano_model in include/harp_hip.h)."""

    def __init__(self, model, device):
        f64 = lambda k: np.asarray(model[k], np.float64)
        vt, sd, pd = f64("v_template").reshape(778, 3), f64("shapedirs").reshape(778, 3, 10), f64("posedirs").reshape(778, 3, 135)
        jr, w, hm = f64("J_regressor").reshape(16, 778), f64("weights").reshape(778, 16), f64("hands_mean").reshape(45)
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)
        self.v_template = up(vt)
        self.shapedirs_T = up(sd.reshape(2334, 10).T)
        self.posedirs_T = up(pd.reshape(2334, 135).T)
        self.posedirs = up(pd.reshape(2334, 135))
        self.J_template = up(jr @ vt)                                   # manolayer.py:188 folded into the shape basis
        self.J_dirs = up(np.einsum("jv,vck->jck", jr, sd).reshape(48, 10))
        self.weights = up(w)
        self.hands_mean = up(hm)
        self.struct = _lib.ManoModel(*[_lib.ptr(getattr(self, n)) for n, _ in _lib.ManoModel._fields_])
        self.device = device


class _ManoLBS(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pose, betas, trans, dm):
        pose, betas, trans = pose.contiguous().float(), betas.contiguous().float(), trans.contiguous().float()
        B = pose.shape[0]
        L = _lib.lib()
        ws = torch.empty(L.harp_lbs_mano_ws_floats(B), dtype=torch.float32, device=pose.device)
        verts = torch.empty(B, 778, 3, dtype=torch.float32, device=pose.device)
        joints = torch.empty(B, 21, 3, dtype=torch.float32, device=pose.device)
        _lib.check(L.harp_lbs_mano_fwd(ctypes.byref(dm.struct), _lib.ptr(pose), _lib.ptr(betas), _lib.ptr(trans), B, _lib.ptr(ws),
                                       _lib.ptr(verts), _lib.ptr(joints), _lib.stream()), "harp_lbs_mano_fwd")
        ctx.save_for_backward(pose, betas, trans, ws)
        ctx.dm = dm
        return verts, joints

    @staticmethod
    def backward(ctx, g_verts, g_joints):
        pose, betas, trans, ws = ctx.saved_tensors
        B = pose.shape[0]
        gv = g_verts.contiguous().float().clone() if g_verts is not None else torch.zeros(B, 778, 3, device=pose.device)
        gj = g_joints.contiguous().float() if g_joints is not None else torch.zeros(B, 21, 3, device=pose.device)
        g_pose, g_betas, g_trans = torch.empty_like(pose), torch.empty_like(betas), torch.empty_like(trans)
        _lib.check(_lib.lib().harp_lbs_mano_bwd(ctypes.byref(ctx.dm.struct), _lib.ptr(pose), _lib.ptr(betas), _lib.ptr(trans), B,
                                                _lib.ptr(ws), _lib.ptr(gv), _lib.ptr(gj), _lib.ptr(g_pose), _lib.ptr(g_betas),
                                                _lib.ptr(g_trans), _lib.stream()), "harp_lbs_mano_bwd")
        return g_pose, g_betas, g_trans, None


class _ChStub:
    """stand-in for chumpy objects when unpickling MANO_RIGHT.pkl without chumpy installed"""
    def __init__(self, *a, **k): pass
    def __setstate__(self, state): self.__dict__.update(state if isinstance(state, dict) else {"x": state})


class _MANOUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.startswith("chumpy"):
            return _ChStub
        return super().find_class(module, name)


def _as_np(v):
    if isinstance(v, _ChStub):
        v = v.__dict__.get("x", v.__dict__.get("r"))
    if hasattr(v, "toarray"):
        v = v.toarray()
    return np.asarray(v)


def load_mano_pkl(path):
    """Best-effort reader of the licensed MANO_RIGHT.pkl (the reference goes through chumpy:
    mano/webuser/smpl_handpca_wrapper_HAND_only.py:22-67). Untested here: the file cannot be shipped."""
    with open(path, "rb") as f:
        d = _MANOUnpickler(f, encoding="latin1").load()
    return dict(v_template=_as_np(d["v_template"]), shapedirs=_as_np(d["shapedirs"]), posedirs=_as_np(d["posedirs"]),
                J_regressor=_as_np(d["J_regressor"]), weights=_as_np(d["weights"]), hands_mean=_as_np(d["hands_mean"]),
                faces=_as_np(d["f"]).astype(np.int64))


class ManoLayer(torch.nn.Module):
    def __init__(self, center_idx=None, flat_hand_mean=True, ncomps=6, side="right", mano_root="mano/models", use_pca=True,
                 root_rot_mode="axisang", joint_rot_mode="axisang", robust_rot=False, model=None, device="cuda"):
        super().__init__()
        if use_pca or root_rot_mode != "axisang" or joint_rot_mode != "axisang" or side != "right" or center_idx is not None:
            raise NotImplementedError("harp_amd.ManoLayer implements the configuration HARP uses: use_pca=False, axis-angle, right hand "
                                      "(utils/hand_model_utils.py:74)")
        if model is None:
            model = load_mano_pkl(os.path.join(mano_root, "MANO_RIGHT.pkl"))
        model = {k: np.asarray(v) for k, v in model.items()}
        if flat_hand_mean:
            model = dict(model, hands_mean=np.zeros(45, np.float32))                                  # manolayer.py:92-93
        self.side, self.use_pca, self.flat_hand_mean, self.ncomps, self.rot = side, use_pca, flat_hand_mean, 45, 3
        self.register_buffer("th_faces", torch.from_numpy(model["faces"].astype(np.int64)))
        self.register_buffer("th_v_template", torch.from_numpy(model["v_template"].astype(np.float32)).unsqueeze(0))
        self._model_np = model
        self._dm = None
        self._dev = torch.device(device)

    def to(self, device):
        self._dev = torch.device(device)
        self._dm = None
        return super().to(device)

    @property
    def device_model(self):
        if self._dm is None:
            self._dm = ManoDeviceModel(self._model_np, self._dev)
        return self._dm

    def forward(self, th_pose_coeffs, th_betas=torch.zeros(1), th_trans=torch.zeros(1), root_palm=torch.Tensor([0]),
                share_betas=torch.Tensor([0]), no_root_rot=False):
        if bool(root_palm) or bool(share_betas) or no_root_rot:
            raise NotImplementedError("root_palm / share_betas / no_root_rot are not used by HARP")
        B = th_pose_coeffs.shape[0]
        dev = th_pose_coeffs.device
        if th_betas is None or th_betas.numel() == 1:
            th_betas = torch.zeros(B, 10, device=dev)                                                 # manolayer.py:176-181 (th_betas buffer = 0)
        if th_trans is None or th_trans.numel() == 1:
            th_trans = torch.zeros(B, 3, device=dev)
        return _ManoLBS.apply(th_pose_coeffs, th_betas, th_trans, self.device_model)
