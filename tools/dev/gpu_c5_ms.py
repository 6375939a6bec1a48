"""C5's per-GPU share (SMPL-X arm mesh 4083 v / 8128 f, 1024^2, 32 frames): per-kernel-group milliseconds (single stream, eager) and the
graph-replayed step; HARP_RASTER_LOOP / HARP_LIB_PATH select variants"""
import sys, os, time; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, bench
eng, focal = bench.build_engine(0, 1, torch.device('cuda'), T=32, img=1024, B=32, kind="arm")
eng.keep_image = False
eng.set_schedule(torch.arange(32).reshape(1, 32))
for _ in range(3):
    eng.step(None, True, True, use_graph=False)
kt = bench.kernel_roofline(eng, 4)
print(os.environ.get("HARP_RASTER_LOOP", "default"), {k.split("(")[0]: round(v, 4) for k, v in kt.items()})
for _ in range(3): eng.step(None, True, True)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(30): eng.step(None, True, True)
torch.cuda.synchronize(); print("step %.3f ms" % ((time.perf_counter() - t) / 30 * 1e3))
nact = __import__("harp_amd.ops", fromlist=["x"]).rasterize_ws_nact(eng.s["ws_c"], eng.B, eng.topo.F, eng.S); print("active super-tiles", nact, "of", 32 * 256)
