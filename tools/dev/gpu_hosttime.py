"""host-side issue time of one eager step vs its GPU time (is the N>1 eager path host-bound?)"""
import sys, os, time; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, bench
eng, _ = bench.build_engine(0, 1, torch.device('cuda'), T=64, img=512, B=32)
sched = torch.stack([(torch.arange(32) + i * 32) % 64 for i in range(4)]).int()
eng.set_schedule(sched)
for g in (False, True):
    for _ in range(5): eng.step(None, True, True, use_graph=g)
    torch.cuda.synchronize()
    n = 15
    t0 = time.perf_counter()
    for _ in range(n): eng.step(None, True, True, use_graph=g)
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'graph={g}: host issue {(t1 - t0) / n * 1e3:.3f} ms/step, total {(t2 - t0) / n * 1e3:.3f} ms/step', flush=True)
