"""save_result / load_result / set_require_grad (utils/file_utils.py:6-37): the same `saved_params[_test].pkl` pickle of
numpy arrays, so checkpoints move both ways between the reference and this implementation."""
import os
import pickle

import torch


def save_result(params, base_output_dir, test=False):
    save_params = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in params.items()}
    with open(os.path.join(base_output_dir, "saved_params" + ("_test" if test else "") + ".pkl"), "wb") as outfile:
        pickle.dump(save_params, outfile)


def load_result(base_output_dir, device="cuda", test=False):
    with open(os.path.join(base_output_dir, "saved_params" + ("_test" if test else "") + ".pkl"), "rb") as infile:
        params = pickle.load(infile)
    for k in params:
        if params[k] is not None:
            params[k] = torch.from_numpy(params[k])
    return set_require_grad(params, device)


def set_require_grad(params, device="cuda"):
    for k in ["trans", "pose", "wrist_pose", "rot", "shape", "verts_disps", "verts_rgb", "texture", "light_positions", "normal_map", "nimble_tex"]:
        if k in params:
            params[k] = torch.nn.Parameter(params[k].to(device) if k == "verts_disps" else params[k], requires_grad=True)
    return params
