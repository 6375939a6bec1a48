"""The data-parallel fit of tests/fit_worker.py in four configurations — {2 ranks (gloo, eager steps), 1 rank (graphs)} x {texel records,
table form} — and the pairwise differences of the fitted texture / normal map: which pair moves when a switch moves."""
import os, sys, tempfile; ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'); sys.path.insert(0, ROOT)
import torch
from tests.test_gpu_dist import _launch
sw = sys.argv[1] if len(sys.argv) > 1 else "texel_records"
tmp = tempfile.mkdtemp()
res = {}
for world in (2, 1):
    for v in (1, 0):
        res[(world, v)] = _launch(world, os.path.join(tmp, f"f{world}{v}.pt"), 0, worker="fit_worker.py", args=[2], env_extra={"HARP_ENG": f"{sw}={v}"})
def diff(a, b, k):
    o, n = a["offsets"][k]; o -= a["opt_lo"]
    d = (a["params"][o:o + n].double() - b["params"][o:o + n].double()).abs()
    return "%s mean %.2e max %.2e frac>1e-3 %.2e" % (k, d.mean(), d.max(), (d > 1e-3).double().mean())
for x, y in (((2, 1), (1, 1)), ((2, 0), (1, 0)), ((2, 1), (2, 0)), ((1, 1), (1, 0))):
    print(f"world {x[0]} {sw}={x[1]}  vs  world {y[0]} {sw}={y[1]}:", diff(res[x], res[y], "texture"), "|", diff(res[x], res[y], "normal_map"))
