"""graph-replayed step time of the bench workload for the library in HARP_LIB_PATH (A/B of kernel variants)"""
import sys, os, time; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, bench
eng, focal = bench.build_engine(0, 1, torch.device('cuda'))
eng.keep_image = False
for kv in filter(None, os.environ.get("HARP_ENG", "").split(",")):      # HARP_ENG="mesh_third=0,camera_first=0": engine switches
    k, v = kv.split("="); setattr(eng, k, type(getattr(eng, k))(int(v)))
eng.set_schedule(torch.arange(256).reshape(-1, 32).int())
res = []
for rep in range(3):
    for _ in range(6): eng.step(None, True, True)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(60): eng.step(None, True, True)
    torch.cuda.synchronize(); res.append((time.perf_counter() - t) / 60 * 1e3)
print(os.path.basename(os.environ.get("HARP_LIB_PATH", "default")), os.environ.get("HARP_ENG", ""), " ".join("%.4f" % r for r in res), "ms/step")
