"""Stand-alone kernel durations of one step: eager, ONE stream (nothing runs next to anything), meant to run under
    rocprofv3 --kernel-trace -d DIR -o run -- python tools/dev/gpu_alone_stats.py ; python tools/rocpd_stats.py DIR/.../*.db
(HARP_ALONE_KIND=arm HARP_ALONE_S=1024 for C5's share)"""
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch, bench
kind, S = os.environ.get("HARP_ALONE_KIND", "hand"), int(os.environ.get("HARP_ALONE_S", "512"))
eng, focal = bench.build_engine(0, 1, torch.device("cuda"), T=32, img=S, B=32, kind=kind)
eng.keep_image = False
eng.overlap = False
for i in range(int(os.environ.get("HARP_ALONE_STEPS", "12"))):
    eng.step(torch.arange(32), True, True, use_graph=False)
torch.cuda.synchronize()
