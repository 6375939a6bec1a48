"""print bench.py's event-timed per-kernel-group milliseconds (single stream, eager) for the library in HARP_LIB_PATH"""
import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, bench
eng, focal = bench.build_engine(0, 1, torch.device('cuda'))
eng.set_schedule(torch.arange(256).reshape(-1, 32))
for _ in range(3):
    eng.step(None, True, True, use_graph=False)
kt = bench.kernel_roofline(eng, 6)
print({k: round(v, 4) for k, v in kt.items()})
