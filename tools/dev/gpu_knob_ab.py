"""A/B of engine switches on the bench workload (graph-replayed steps), several captures per configuration:
   python tools/dev/gpu_knob_ab.py "fold_step=0" "fold_step=1" "fold_step=1,fused_terms=0" ...   (legacy form: NAME v0 v1 ...)"""
import sys, os, time; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, bench, statistics
args = sys.argv[1:]
if args and "=" not in args[0]:
    args = ["%s=%s" % (args[0], v) for v in args[1:]]
cfgs = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split(",") if kv) for a in args]
kind, img = os.environ.get("HARP_TL_KIND", "hand"), int(os.environ.get("HARP_TL_IMG", "512"))      # C5: HARP_TL_KIND=arm HARP_TL_IMG=1024
eng, focal = bench.build_engine(0, 1, torch.device('cuda'), kind=kind, img=img)
eng.keep_image = False
eng.set_schedule(torch.arange(256).reshape(-1, 32).int())
defaults = {k: getattr(eng, k) for c in cfgs for k in c}
def run(cfg, n=200):
    for k, v in defaults.items(): setattr(eng, k, v)
    for k, v in cfg.items(): setattr(eng, k, type(defaults[k])(v))
    eng._graphs = {}
    for _ in range(20): eng.step(None, True, True)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): eng.step(None, True, True)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
res = {i: [] for i in range(len(cfgs))}
for rep in range(int(os.environ.get("REPS", "5"))):
    for i, c in enumerate(cfgs):
        res[i].append(run(c))
for i, c in enumerate(cfgs):
    r = res[i]
    print("%-44s median %.4f  min %.4f  max %.4f   %s" % (args[i], statistics.median(r), min(r), max(r), " ".join("%.4f" % x for x in r)), flush=True)
