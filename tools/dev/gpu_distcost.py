"""cost of the N>1 step variants on a 1-rank RCCL group (upper bound of the fixed collective overhead)"""
import sys, os, time; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29544", RANK="0", WORLD_SIZE="1")
import torch, torch.distributed as dist, bench
dev = torch.device('cuda', 0)
dist.init_process_group("nccl", device_id=dev); dist.barrier(); torch.cuda.synchronize()
eng, _ = bench.build_engine(0, 1, dev, T=64, img=512, B=32)
sched = torch.stack([(torch.arange(32) + i * 32) % 64 for i in range(4)]).int()
eng.set_schedule(sched)
def t(n=60):
    for _ in range(6): eng.step(None, True, True, use_graph=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): eng.step(None, True, True, use_graph=True)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print('single GPU graph                 ', round(t(), 4))
eng.force_allreduce = True
for ov in (True, False):
    eng.overlap_allreduce = ov; eng._graphs = {}
    print(f'eager + RCCL, early overlap={ov}   ', round(t(), 4))
eng.graph_collectives = True; eng._graphs = {}
print('graph-captured RCCL (one bucket) ', round(t(), 4))
eng.graph_collectives = False; eng.overlap_allreduce = False; eng._graphs = {}
for _ in range(6): eng.step(None, True, True, use_graph=True)
torch.cuda.synchronize(); n = 20; t0 = time.perf_counter()
for _ in range(n): eng.step(None, True, True, use_graph=True)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f'eager + RCCL single bucket: host issue {(t1 - t0) / n * 1e3:.3f} ms/step, total {(t2 - t0) / n * 1e3:.3f} ms/step')
dist.destroy_process_group()
