"""step time with / without the second stream, eager and graph (is the fork / join worth it?)"""
import sys, os, time; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, bench
eng, _ = bench.build_engine(0, 1, torch.device('cuda'), T=64, img=512, B=32)
sched = torch.stack([(torch.arange(32) + i * 32) % 64 for i in range(4)]).int()
eng.set_schedule(sched)
def t(graph, n=60):
    for _ in range(5): eng.step(None, True, True, use_graph=graph)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): eng.step(None, True, True, use_graph=graph)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for overlap in (True, False):
    for early in (True, False):
        eng.overlap, eng.early_terms = overlap, early
        eng._graphs = {}
        print(f'overlap={overlap} early_terms={early}: eager {t(False):.3f} ms  graph {t(True):.3f} ms', flush=True)
