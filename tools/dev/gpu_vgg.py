"""Cost of the optional perceptual term at the bench configuration (B=32, 512^2): ms/step of the full stage with the term off /
on (float32 MFMA, cached target features) / on (float32, uncached) / on (bf16 split, cached / uncached).  Random filters (timing only)."""
import sys, time
import torch
sys.path.insert(0, ".")
import bench
from harp_amd.model.vgg import Vgg16Features


def time_steps(eng, n=5):
    for _ in range(2):
        eng.step(None, True, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        eng.step(None, True, True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
eng, _ = bench.build_engine(0, 1, torch.device("cuda:0"), T=T)
eng.set_schedule(torch.arange(T).reshape(-1, eng.B))
print("term off            %8.2f ms/step" % time_steps(eng, 20), flush=True)
vgg = Vgg16Features(layers_weights=[1, 1 / 16, 1 / 8, 1 / 4, 1], weights="random")
for name, kw in (("f32 bounded", dict()), ("f32 full, taps cached", dict(bounded=False)), ("f32 uncached", dict(cache_bytes=0)),
                 ("bf16x3 bounded", dict(precision=1)), ("bf16x3 full, taps cached", dict(precision=1, bounded=False))):
    t0 = time.perf_counter()
    eng.set_perceptual(vgg, **kw)
    torch.cuda.synchronize()
    t_set = time.perf_counter() - t0
    ms = time_steps(eng)
    if eng._vgg_bound is not None:
        frac = [float(b[2].float().sum()) * b[6] ** 2 / (b[0].shape[0] * float(eng.S >> lv) ** 2) for lv, b in enumerate(eng._vgg_bound)]
        name += " (tiles %s)" % "/".join("%.2f" % f for f in frac)
    print("%-26s %8.2f ms/step  (set_perceptual %.2f s, peak mem %.1f GB, vgg loss %.5f)" %
          (name, ms, t_set, torch.cuda.max_memory_allocated() / 2**30, eng.losses()["vgg"]), flush=True)
