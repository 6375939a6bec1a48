"""single-stream, event-timed kernel groups of the bench workload for the library in HARP_LIB_PATH (ablation variants: -DRASTER_ABLATE=n)"""
import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, bench
eng, focal = bench.build_engine(0, 1, torch.device('cuda'))
eng.keep_image = False
eng.set_schedule(torch.arange(256).reshape(-1, 32).int())
for _ in range(4): eng.step(None, True, True, use_graph=False)
r = bench.kernel_roofline(eng, 10)
print(os.path.basename(os.environ.get("HARP_LIB_PATH", "default")), {k.split("(")[0]: round(v, 4) for k, v in r.items()})
