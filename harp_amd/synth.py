"""Synthetic, licence-free stand-ins for the assets the reference needs but cannot ship
(MANO_RIGHT.pkl, METRO per-frame fits, captured images): SURVEY.md §8(d) "Synthetic inputs".

Deterministic (numpy default_rng(seed)); used by bench.py, __graft_entry__.smoke() and the tests.
Topology / UVs / UV mask are the reference's real template data (harp_amd/assets, built by
tools/make_assets.py), so problem sizes are the real ones: hand 778->3093 v, 1538->6152 f, 3327 uv.
"""
import os

import numpy as np
import torch

from .topology import (csr_from_pairs, normal_consistency_pairs, subdivide_topology, unique_edges)

_ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")

MANO_TIPS = [745, 317, 444, 556, 673]            # thumb, index, middle, ring, pinky (manolayer.py:270)
# MANO kinematic order: joints 1-3 index, 4-6 middle, 7-9 pinky, 10-12 ring, 13-15 thumb
_FINGER_OF_CHAIN = [1, 2, 4, 3, 0]               # chain c (joints 3c+1..3c+3) -> index into MANO_TIPS


def load_template(name="hand"):
    d = np.load(os.path.join(_ASSETS, f"{name}_template.npz"))
    return {k: d[k] for k in d.files}


def build_topology(faces0, n_verts0):
    """All static int tables for one template (numpy). Keys:
    faces0 (F0,3), edges0 (E0,2), faces (F,3) subdivided, edges (E,2) of the subdivided mesh,
    nbr_off/nbr_idx (vertex->neighbour CSR), vf_off/vf_idx (vertex->(face*3+corner) CSR),
    nc_pairs (P,4), vp_off/vp_idx (vertex->(pair*4+role) CSR), sub_off/sub_idx (base-vertex -> child midpoint CSR, for subdivide backward)."""
    faces0 = np.asarray(faces0, np.int64)
    edges0, faces = subdivide_topology(faces0, n_verts0)
    V = n_verts0 + len(edges0)
    edges, _ = unique_edges(faces, V)
    rows = np.concatenate([edges[:, 0], edges[:, 1]])
    cols = np.concatenate([edges[:, 1], edges[:, 0]])
    nbr_off, nbr_idx = csr_from_pairs(rows, cols, V)
    fc = np.arange(faces.size)
    vf_off, vf_idx = csr_from_pairs(faces.reshape(-1), fc, V)
    nc_pairs = normal_consistency_pairs(faces, V)
    vp_off, vp_idx = csr_from_pairs(nc_pairs.reshape(-1), np.arange(nc_pairs.size), V)   # vertex -> pair*4+role
    srow = np.concatenate([edges0[:, 0], edges0[:, 1]])
    scol = np.concatenate([np.arange(len(edges0)), np.arange(len(edges0))]) + n_verts0
    sub_off, sub_idx = csr_from_pairs(srow, scol, n_verts0)
    return dict(faces0=faces0.astype(np.int32), edges0=edges0.astype(np.int32), faces=faces.astype(np.int32),
                edges=edges.astype(np.int32), nbr_off=nbr_off, nbr_idx=nbr_idx, vf_off=vf_off, vf_idx=vf_idx,
                nc_pairs=nc_pairs, vp_off=vp_off, vp_idx=vp_idx, sub_off=sub_off, sub_idx=sub_idx, n_verts0=n_verts0, n_verts=V)


def build_raw_topology(faces0, n_verts0):
    """The same tables for the UN-subdivided template: `prepare_mesh(..., mesh_subdivider=None)` (utils/visualize.py:51-56) and
    BASELINE config C1 (raw 778-vertex / 1538-face MANO mesh).  V == V0, no midpoints (E0 = 0, empty subdivision tables)."""
    faces = np.asarray(faces0, np.int64)
    V = int(n_verts0)
    edges, _ = unique_edges(faces, V)
    rows = np.concatenate([edges[:, 0], edges[:, 1]])
    cols = np.concatenate([edges[:, 1], edges[:, 0]])
    nbr_off, nbr_idx = csr_from_pairs(rows, cols, V)
    vf_off, vf_idx = csr_from_pairs(faces.reshape(-1), np.arange(faces.size), V)
    nc_pairs = normal_consistency_pairs(faces, V)
    vp_off, vp_idx = csr_from_pairs(nc_pairs.reshape(-1), np.arange(nc_pairs.size), V)
    return dict(faces0=faces.astype(np.int32), edges0=np.zeros((0, 2), np.int32), faces=faces.astype(np.int32), edges=edges.astype(np.int32),
                nbr_off=nbr_off, nbr_idx=nbr_idx, vf_off=vf_off, vf_idx=vf_idx, nc_pairs=nc_pairs, vp_off=vp_off, vp_idx=vp_idx,
                sub_off=np.zeros(V + 1, np.int32), sub_idx=np.zeros(0, np.int32), n_verts0=V, n_verts=V)


def make_mano_model(template=None, seed=0):
    """A MANO-shaped LBS model on the real MANO topology with synthetic blend shapes (float32 numpy):
    v_template (778,3) m, shapedirs (778,3,10), posedirs (778,3,135), J_regressor (16,778),
    weights (778,16), hands_mean (45,), faces (1538,3)."""
    t = template or load_template("hand")
    rng = np.random.default_rng(seed)
    v = t["base_verts"].astype(np.float64)
    root = v[v[:, 0] > v[:, 0].max() - 0.02].mean(0)
    pts = [root]
    for c in range(5):
        tip = v[MANO_TIPS[_FINGER_OF_CHAIN[c]]]
        for fr in (0.50, 0.68, 0.85):
            pts.append(root + (tip - root) * fr)
    pts = np.stack(pts)                                             # (16,3)
    d2 = ((v[None] - pts[:, None]) ** 2).sum(-1)                    # (16,778)
    jr = np.exp(-d2 / (2 * 0.012 ** 2)) + 1e-12
    jr /= jr.sum(1, keepdims=True)
    w = np.exp(-d2.T / (2 * 0.010 ** 2)) + 1e-9
    w /= w.sum(1, keepdims=True)
    return dict(v_template=v.astype(np.float32),
                shapedirs=(rng.standard_normal((778, 3, 10)) * 1e-3).astype(np.float32),
                posedirs=(rng.standard_normal((778, 3, 135)) * 1e-3).astype(np.float32),
                J_regressor=jr.astype(np.float32), weights=w.astype(np.float32),
                hands_mean=(rng.standard_normal(45) * 0.05).astype(np.float32),
                faces=t["faces0"].astype(np.int64))


def make_sequence(model, T, S, seed=0, focal=None):
    """Per-frame METRO-style inputs (the dict `init_params` consumes, optimize_sequence.py:181-250):
    pose (T,45), rot (T,3), trans (T,3), shape (T,10), cam (T,3), joints (T,21,3) mm filled by caller."""
    rng = np.random.default_rng(seed + 1000)
    focal = focal if focal is not None else 1000.0 * S / 224.0      # utils/config_utils.py:12
    k = np.hanning(9)
    k /= k.sum()

    def smooth(x):
        if x.shape[0] < 9:
            return x
        pad = np.pad(x, ((4, 4), (0, 0)), mode="edge")
        return np.stack([np.convolve(pad[:, i], k, mode="valid") for i in range(x.shape[1])], 1)

    pose = smooth(rng.standard_normal((T, 45)) * 0.2)
    rot = smooth(rng.standard_normal((T, 3)) * 0.3) + np.array([0.3, -0.4, 0.2])
    shape = np.tile(rng.standard_normal((1, 10)) * 0.5, (T, 1)) + rng.standard_normal((T, 10)) * 0.05
    z = rng.uniform(0.85, 1.25, (T, 1))                             # depth in metres
    s = 2.0 * focal / (S * z)                                       # cam[:,0]: t_z = 2f/(S*s)
    # keep the posed hand roughly centred: centroid after the root rotation about the root joint
    c0 = model["v_template"].mean(0).astype(np.float64)
    rj = (model["J_regressor"][0].astype(np.float64)[:, None] * model["v_template"]).sum(0)
    ang = np.linalg.norm(rot, axis=1, keepdims=True) + 1e-12
    ax = rot / ang
    d = (c0 - rj)[None]
    cr = d * np.cos(ang) + np.cross(ax, d) * np.sin(ang) + ax * (ax * d).sum(1, keepdims=True) * (1 - np.cos(ang)) + rj
    txy = rng.uniform(-0.015, 0.015, (T, 2)) - cr[:, :2]
    cam = np.concatenate([s, txy], 1)
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return dict(pose=f32(pose), rot=f32(rot), trans=torch.zeros(T, 3), shape=f32(shape), cam=f32(cam)), float(focal)


SMPLX_PARENTS = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 15, 15, 15, 20, 25, 26, 20, 28, 29, 20, 31, 32,
                 20, 34, 35, 20, 37, 38, 21, 40, 41, 21, 43, 44, 21, 46, 47, 21, 49, 50, 21, 52, 53]


def make_smplx_arm_model(template=None, seed=0):
    """An SMPL-X-right-arm-shaped LBS model on the real arm topology (1026 v / 2032 f) with synthetic blend shapes, already
    arm-sliced with the joint regressor folded in (see harp_amd/hand_models_harp/body_models.py): v_template (1026,3) m,
    shapedirs (1026,3,20), posedirs (486, 3078), J_template (55,3), J_shapedirs (55,3,20), weights (1026,55), pose_mean (165,),
    parents (55,), tip_verts (5,), faces (2032,3)."""
    t = template or load_template("arm")
    corr = np.load(os.path.join(_ASSETS, "arm_corr.npz"))
    rng = np.random.default_rng(seed + 17)
    v = t["base_verts"].astype(np.float64)
    mano_from_arm = corr["mano_vert_from_arm"].astype(np.int64)
    tip_verts = mano_from_arm[MANO_TIPS]                                   # thumb, index, middle, ring, pinky (arm-local ids)
    hand = v[mano_from_arm]
    axis = v[v[:, 0] > np.quantile(v[:, 0], 0.98)].mean(0) - hand.mean(0)  # hand -> upper arm direction
    L = np.linalg.norm(axis)
    axis /= L
    wrist = hand[np.argsort((hand - hand.mean(0)) @ axis)[-30:]].mean(0)   # hand vertices closest to the forearm
    J = np.zeros((55, 3))
    J[21] = wrist
    J[19] = wrist + axis * 0.55 * L                                        # right elbow
    J[17] = wrist + axis * 1.05 * L                                        # right shoulder
    J[14] = J[17] + axis * 0.08
    J[9], J[6], J[3], J[0] = J[14] + axis * 0.05, J[14] + axis * 0.10, J[14] + axis * 0.15, J[14] + axis * 0.20
    others = [j for j in range(55) if not J[j].any()]
    J[others] = J[0] + rng.standard_normal((len(others), 3)) * 0.05
    # SMPL-X right-hand joints: index 40-42, middle 43-45, pinky 46-48, ring 49-51, thumb 52-54
    for base, tip in ((40, 1), (43, 2), (46, 4), (49, 3), (52, 0)):
        for k, fr in enumerate((0.50, 0.68, 0.85)):
            J[base + k] = wrist + (v[tip_verts[tip]] - wrist) * fr
    active = [17, 19, 21] + list(range(40, 55))
    d2 = ((v[:, None] - J[None, active]) ** 2).sum(-1)
    w_act = np.exp(-d2 / (2 * 0.012 ** 2)) + 1e-9
    w_act[:, :3] = np.exp(-d2[:, :3] / (2 * 0.05 ** 2)) + 1e-9             # broader influence of the arm joints
    w = np.zeros((len(v), 55))
    w[:, active] = w_act
    w /= w.sum(1, keepdims=True)
    pose_mean = np.zeros(165)
    pose_mean[120:165] = rng.standard_normal(45) * 0.05                    # flat_hand_mean=False: right-hand mean pose
    return dict(v_template=v.astype(np.float32), shapedirs=(rng.standard_normal((len(v), 3, 20)) * 1e-3).astype(np.float32),
                posedirs=(rng.standard_normal((486, len(v) * 3)) * 1e-3).astype(np.float32), J_template=J.astype(np.float32),
                J_shapedirs=(rng.standard_normal((55, 3, 20)) * 1e-3).astype(np.float32), weights=w.astype(np.float32),
                pose_mean=pose_mean.astype(np.float32), parents=np.asarray(SMPLX_PARENTS, np.int32),
                tip_verts=tip_verts.astype(np.int32), faces=t["faces0"].astype(np.int64))
