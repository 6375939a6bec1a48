"""a few steps of the bench step with the perceptual term on (float32, bounded) — run under rocprofv3 --kernel-trace for per-launch durations"""
import sys, torch
sys.path.insert(0, ".")
import bench
from harp_amd.model.vgg import Vgg16Features
prec = int(sys.argv[1]) if len(sys.argv) > 1 else 0
eng, _ = bench.build_engine(0, 1, torch.device("cuda:0"), T=64)
eng.set_schedule(torch.arange(64).reshape(-1, eng.B))
eng.set_perceptual(Vgg16Features(layers_weights=[1, 1 / 16, 1 / 8, 1 / 4, 1], weights="random"), precision=prec)
for _ in range(6):
    eng.step(None, True, True)
torch.cuda.synchronize()
