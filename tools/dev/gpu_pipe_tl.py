"""graph replays of the bench workload with FitEngine attributes set from HARP_ENG (e.g. "pipelined=1,mesh_third=0"): for rocprofv3 timelines"""
import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, bench
kind, img = os.environ.get("HARP_TL_KIND", "hand"), int(os.environ.get("HARP_TL_IMG", "512"))      # C5: HARP_TL_KIND=arm HARP_TL_IMG=1024
eng, focal = bench.build_engine(0, 1, torch.device('cuda'), kind=kind, img=img)
eng.keep_image = False
for kv in filter(None, os.environ.get("HARP_ENG", "").split(",")):
    k, v = kv.split("="); setattr(eng, k, type(getattr(eng, k))(int(v)))
eng.set_schedule(torch.arange(256).reshape(-1, 32).int())
c, a = (os.environ.get("HARP_TL_STAGE", "11")[0] == "1"), (os.environ.get("HARP_TL_STAGE", "11")[1] == "1")      # HARP_TL_STAGE=10: geometry only, 01: appearance only
for _ in range(30): eng.step(None, c, a)
torch.cuda.synchronize()
