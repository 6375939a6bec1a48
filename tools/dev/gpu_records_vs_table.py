"""One step's texture / normal-map gradient with texel records and with the table form, element by element: where do the two differ by
more than float rounding, and would Adam (eps 1e-8) care?"""
import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from tests._scene import make_fit_case
case = make_fit_case("hand", T=4, S=96, B=4, seed=6, device="cuda")
eng = case["eng"]
eng.keep_image = False; eng.auto_draw = False; eng.draw_texture_offsets(); eng.set_lr(0.0, 0.0)
eng.set_schedule(torch.arange(4).reshape(1, 4).int())
out = {}
for rec in (False, True):
    eng.texel_records = rec
    eng.step(None, True, True, use_graph=False); torch.cuda.synchronize()
    out[rec] = {k: eng.grads[k].double().clone().flatten() for k in ("texture", "normal_map")}
    out[rec]["g_nmap_n"] = (eng._tacc[1].clone() if False else eng.g_nmap_n.double().clone().flatten())
for k in ("texture", "normal_map"):
    a, b = out[True][k], out[False][k]
    nz = (a != 0) | (b != 0)
    d = (a - b).abs()
    print(k, "nonzero entries", int(nz.sum()), "| zero in table only", int(((b == 0) & (a != 0)).sum()), "| zero in records only", int(((a == 0) & (b != 0)).sum()),
          "| |diff| > 1e-9:", int((d > 1e-9).sum()), "> 1e-8:", int((d > 1e-8).sum()), "| max |diff| %.3e at |g| %.3e" % (d.max().item(), b[d.argmax()].abs().item()),
          "| median |g| %.2e" % b[nz].abs().median().item(), "| sign differs (both nonzero):", int(((a * b) < 0).sum()))
    big = d > 1e-9
    if big.any():
        i = torch.nonzero(big).flatten()[:12]
        print("   examples (records, table):", [(float("%.3e" % a[j]), float("%.3e" % b[j])) for j in i])
