"""which part of the shader backward's GEOMETRY gradient (appearance-only stage) deviates from the fp64 oracle?"""
import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from tests._scene import ORACLE_KEYS, engine_eval, make_fit_case, mask_ambiguous_pixels, oracle_step, rel

def run(tag, shadow=True, const_tex=False, flat_nmap=False, S=128, T=2, B=2, keep=True, coarse=False, seed=3, fused=True, overlap=True, mask=True):
    case = make_fit_case("hand", T=T, S=S, B=B, seed=seed, device="cuda", self_shadow=shadow)
    eng = case["eng"]
    with torch.no_grad():
        if const_tex: eng.params["texture"].fill_(0.6)
        if flat_nmap: eng.params["normal_map"].copy_(torch.tensor([0., 0., 1.], device="cuda").repeat(1, 512, 512, 1))
    frac = mask_ambiguous_pixels(case) if mask else 0.0
    eng.fused_chain = fused and eng.fused_chain
    eng.overlap = overlap
    eng.keep_image = keep
    eng.draw_texture_offsets()
    fid = torch.arange(B)
    lv = engine_eval(case, fid, coarse=coarse, app=True)
    P, loss, total, aux, _ = oracle_step(case, fid, coarse=coarse, app=True)
    out = {k: rel(eng.grads[k].cpu().double(), P[k].grad) for k in ("pose", "cam", "shape", "verts_disps", "rot", "trans", "texture", "normal_map", "light_positions", "amb_ratio") if P[k].grad is not None and P[k].grad.abs().max() > 0}
    print(tag, f"amb {frac:.4f}", {k: f"{v:.1e}" for k, v in out.items()}, flush=True)
    return case, P


run("S=256 noshadow          ", shadow=False, S=256)
run("S=256 shadow            ", shadow=True, S=256)
run("S=256 shadow unfused    ", shadow=True, S=256, fused=False)
run("S=256 shadow 1 stream   ", shadow=True, S=256, overlap=False)
run("S=256 shadow nomask     ", shadow=True, S=256, mask=False)
run("S=192 shadow            ", shadow=True, S=192)
run("S=256 shadow seed 7     ", shadow=True, S=256, seed=7)
run("S=256 shadow consttex fl", shadow=True, S=256, const_tex=True, flat_nmap=True)
run("S=128 T=25 B=18 full    ", shadow=True, S=128, T=25, B=18, coarse=True, seed=1)
