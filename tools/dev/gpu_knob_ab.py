"""A/B of a numeric engine knob on the bench workload (graph-replayed steps): python tools/dev/gpu_knob_ab.py NAME v0 v1 ..."""
import sys, os, time; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, bench
name, vals = sys.argv[1], [int(v) for v in sys.argv[2:]]
eng, focal = bench.build_engine(0, 1, torch.device('cuda'))
eng.keep_image = False
eng.set_schedule(torch.arange(256).reshape(-1, 32).int())
def run(val, n=60):
    setattr(eng, name, val); eng._graphs = {}
    for _ in range(6): eng.step(None, True, True)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): eng.step(None, True, True)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
for rep in range(3):
    print(name, {v: round(run(v), 4) for v in vals}, flush=True)
