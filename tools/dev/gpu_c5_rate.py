import sys, os, time; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, bench
dev = torch.device('cuda')
for T in (32, 256):
    e = bench.build_engine(0, 1, dev, T=T, img=1024, B=32, kind="arm")[0]
    e.keep_image = False
    for steps, warm in ((40, 6), (40, 6), (200, 20), (40, 6)):
        r = bench._graph_rate(e, steps, warm)
        print("T", T, "steps", steps, "warmup", warm, "ms/step %.4f" % r["ms_per_step"], flush=True)
    del e; torch.cuda.empty_cache()
