"""load_hand_model (utils/hand_model_utils.py:11-82) for the model types on the hot path: the UV layout from the template OBJ
(`pytorch3d.io.load_obj(...).verts_uvs / faces.textures_idx`, :58-60) and the hand layer — METRO-compatible MANO (:74) or the
SMPL-X arm (:66-70).  'html' / 'nimble' are out of scope."""
import numpy as np
import torch


def load_obj_uvs(path):
    """(verts_uvs (Nt,2) float32, faces_uvs (F,3) int64) of a triangulated OBJ: the `vt` lines and the second index of every `f` corner
    — what pytorch3d.io.load_obj returns as properties.verts_uvs and faces.textures_idx.  Like load_obj: negative (relative) indices are
    resolved against the FINAL number of `vt` lines (after the whole file is read — an OBJ that interleaves `vt` and `f` lines therefore
    reads the same as through load_obj), and a face whose corners carry no texture index ('f v' / 'f v//vn') gets -1."""
    vt, ft, where = [], [], []
    with open(path) as f:
        for line in f:
            p = line.split()
            if not p:
                continue
            if p[0] == "vt":
                vt.append([float(x) for x in p[1:3]])
            elif p[0] == "f":
                if len(p) != 4:
                    raise ValueError(f"{path}: only triangulated OBJ templates are supported")
                row = []
                for q in p[1:4]:
                    parts = q.split("/")
                    row.append(int(parts[1]) if (len(parts) >= 2 and parts[1] != "") else None)
                ft.append(row)
                where.append(line.strip())
    n, out = len(vt), []
    for row, line in zip(ft, where):
        if None in row:
            out.append([-1, -1, -1])
            continue
        for i in row:
            if i == 0 or i > n or -i > n:
                raise ValueError(f"{path}: texture index {i} out of range in {line!r}")
        out.append([i - 1 if i > 0 else n + i for i in row])
    return torch.tensor(np.asarray(vt, np.float32).reshape(-1, 2)), torch.tensor(np.asarray(out, np.int64).reshape(-1, 3))


def load_hand_model(config_dict):
    """-> (hand_layer, VERTS_UVS (1,Nt,2), FACES_UVS (1,F,3), VERTS_COLOR) like the reference"""
    device = config_dict.get("device", "cuda")
    if config_dict["model_type"] != "harp":
        raise NotImplementedError("model_type 'html' / 'nimble' are out of scope (SURVEY.md §2)")
    verts_uvs, faces_uvs = load_obj_uvs(config_dict["MANO_TEMPLATE"])
    if (faces_uvs < 0).any():                # the shaders index verts_uvs[faces_uvs] on the device: a face without UVs cannot be textured
        raise ValueError(f"{config_dict['MANO_TEMPLATE']}: {int((faces_uvs < 0).any(1).sum())} faces carry no texture indices")
    VERTS_COLOR = None
    if config_dict["use_arm"]:
        from ..hand_models_harp import body_models
        hand_layer = body_models.create(config_dict.get("smplx_model_folder", "hand_models/smplx/models/"), model_type="smplxarm",
                                        num_betas=10, num_expression_coeffs=10, device=device)
    else:
        from ..manopth.manolayer import ManoLayer
        from .opt_utils import get_mano_vert_colors
        hand_layer = ManoLayer(mano_root=config_dict.get("mano_root", "mano/models"), flat_hand_mean=False, use_pca=False, device=device)
        VERTS_COLOR = get_mano_vert_colors(hand_layer)
    return hand_layer, verts_uvs.unsqueeze(0), faces_uvs.unsqueeze(0), VERTS_COLOR
