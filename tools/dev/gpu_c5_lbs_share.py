"""share of the kinematic-tree LBS kernels in a C5 step (arm mesh, 1024x1024, 32 frames): run under rocprofv3 --kernel-trace --stats"""
import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import time, torch, bench
e = bench.build_engine(0, 1, torch.device('cuda'), T=32, img=1024, B=32, kind="arm")[0]
e.keep_image = False
e.set_schedule(torch.arange(32).reshape(1, 32).int())
for _ in range(4): e.step(None, True, True)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(20): e.step(None, True, True)
torch.cuda.synchronize(); print("C5 per-GPU step ms", (time.perf_counter() - t) / 20 * 1e3)
