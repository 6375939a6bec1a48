"""What the side-stream work costs the step: graph-replayed step time of the bench workload with groups of side kernels left out
(results are then WRONG; timing only)."""
import sys, os, time; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, bench
eng, focal = bench.build_engine(0, 1, torch.device('cuda'))
eng.keep_image = False
eng.set_schedule(torch.arange(256).reshape(-1, 32).int())
def run(n=80):
    eng._graphs = {}
    for _ in range(6): eng.step(None, True, True)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): eng.step(None, True, True)
    torch.cuda.synchronize(); return round((time.perf_counter() - t) / n * 1e3, 4)
orig_tex = eng._texture_terms
for rep in range(2):
    eng.set_disabled_terms(()); eng._texture_terms = orig_tex
    print("baseline", run(), flush=True)
    eng.set_disabled_terms(("laplacian", "normal", "arap"))
    print("no mesh_reg", run(), flush=True)
    eng.set_disabled_terms(("laplacian", "normal", "arap", "kps_anchor", "vert_disp_reg"))
    print("no mesh_reg, kps, disp_reg", run(), flush=True)
    eng._texture_terms = lambda wp, lp: None
    print("... and no texture regularisers", run(), flush=True)
