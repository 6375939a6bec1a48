import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from tests._scene import ORACLE_KEYS, make_fit_case, oracle_inputs, oracle_step, rel
graph = int(sys.argv[1]) if len(sys.argv) > 1 else 1
case = make_fit_case("hand", T=3, S=128, B=2, seed=4, device="cuda")
eng = case["eng"]; eng.keep_image = False
P, model, targets = oracle_inputs(case)
p0 = {k: P[k].detach().clone() for k in ORACLE_KEYS}
opt_c = torch.optim.Adam([{"params": [P["pose"], P["cam"]], "lr": 1e-3}, {"params": [P["verts_disps"], P["shape"]], "lr": 1e-3}])
opt_a = torch.optim.Adam([P["light_positions"], P["amb_ratio"], P["texture"], P["normal_map"]], lr=1e-2)
eng.auto_draw = True
for it in range(10):
    fid = torch.tensor([it % 3, (it + 1) % 3])
    eng.step(fid, True, True, use_graph=bool(graph) and it > 0)
    torch.cuda.synchronize()
    g_h = {k: eng.grads[k].cpu().double().clone() for k in ("pose", "cam", "shape")}
    opt_c.zero_grad(); opt_a.zero_grad()
    oracle_step(case, fid, P=P, model=model, targets=targets)
    gr = {k: rel(g_h[k], P[k].grad) for k in g_h}
    opt_c.step(); opt_a.step()
    d = (eng.params["pose"].cpu().double() - P["pose"].detach()).abs()
    i = int(d.argmax())
    print(it, "grad rel", {k: f"{v:.1e}" for k, v in gr.items()}, "pose max|d| %.2e at %s" % (d.max().item(), divmod(i, 45)),
          "g_hip %.3e g_ref %.3e" % (g_h["pose"].flatten()[i].item(), P["pose"].grad.flatten()[i].item()), "rel pose %.2e" % rel(eng.params["pose"].cpu().double(), P["pose"].detach()), flush=True)
