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


def oracle_params(sc, source):
    """dict of leaf tensors (requires_grad) cloned from `source` (engine.params or any dict of tensors)."""
    keys = ("pose", "cam", "verts_disps", "shape", "light_positions", "amb_ratio", "texture", "normal_map", "rot", "trans")
    P = {k: source[k].detach().cpu().clone().requires_grad_() for k in keys}
    P.update(verts_uvs=torch.from_numpy(sc["tpl"]["verts_uvs"]), faces_uvs=torch.from_numpy(sc["tpl"]["faces_uvs"]).long(),
             uv_mask=sc["uv_mask"], init_joints=sc["seq"]["joints"])
    return P


def rel(a, b):
    return ((a - b).norm() / (b.norm() + 1e-30)).item()
