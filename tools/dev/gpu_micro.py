import sys, time, faulthandler; sys.path.insert(0,'.'); faulthandler.enable()
import torch, bench
mode = sys.argv[1]
eng, focal = bench.build_engine(0, 1, torch.device('cuda'), T=32, img=512, B=32)
sched = torch.arange(32, dtype=torch.int32, device='cuda')
def run(n, graph):
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): eng.step(sched, True, True, use_graph=graph)
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
if mode == 'eager':
    print('micro eager ms', run(3, False), run(20, False))
elif mode == 'graph_nolaneoverlap':
    eng._inner_overlap = False
    print('graph ms', run(3, True), run(20, True))
elif mode == 'graph':
    print('graph ms', run(3, True), run(20, True))
elif mode == 'nomicro':
    eng.micro = 1
    print('nomicro graph ms', run(3, True), run(20, True))
print(eng.losses())
