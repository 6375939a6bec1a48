"""C5 step time (arm mesh, 1024^2, 32 frames) for the library in HARP_LIB_PATH: python tools/dev/gpu_c5_variants.py"""
import sys, os, time; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, bench
e = bench.build_engine(0, 1, torch.device('cuda'), T=32, img=1024, B=32, kind="arm")[0]
e.keep_image = False
r = [bench._graph_rate(e, 60, 100 if i == 0 else 10)["ms_per_step"] for i in range(3)]
print(os.path.basename(os.environ.get("HARP_LIB_PATH", "default")), " ".join("%.4f" % x for x in r), "ms/step")
