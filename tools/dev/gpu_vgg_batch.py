"""perceptual term (f32, bounded) at B = 32 / 16 / 8 frames per step: ms per step minus the step without the term — how much of the term's
time is per-launch tail (19 dependent convolution launches, each a few rounds of workgroups over the chip)"""
import sys, time
import torch
sys.path.insert(0, ".")
import bench
from harp_amd.model.vgg import Vgg16Features

def time_steps(eng, n=5):
    for _ in range(2):
        eng.step(None, True, True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        eng.step(None, True, True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

vgg = Vgg16Features(layers_weights=[1, 1 / 16, 1 / 8, 1 / 4, 1], weights="random")
for B in (tuple(int(x) for x in sys.argv[1:]) or (32, 16, 8)):
    eng, _ = bench.build_engine(0, 1, torch.device("cuda:0"), T=64, B=B)
    eng.set_schedule(torch.arange(64).reshape(-1, B))
    off = time_steps(eng, 20)
    for prec in (0, 1):
        eng.set_perceptual(vgg, precision=prec)
        on = time_steps(eng)
        print("B %2d prec %d: step %.3f ms, term on %.2f ms -> term %.2f ms = %.3f ms per frame" % (B, prec, off, on, on - off, (on - off) / B), flush=True)
    del eng; torch.cuda.empty_cache()
