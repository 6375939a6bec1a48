"""a few graph-replayed steps of the bench workload with the perceptual term on (for rocprofv3 --kernel-trace --stats): python tools/dev/gpu_vgg_trace.py [precision]"""
import sys
import torch
sys.path.insert(0, ".")
import bench
from harp_amd.model.vgg import Vgg16Features
prec = int(sys.argv[1]) if len(sys.argv) > 1 else 0
eng, _ = bench.build_engine(0, 1, torch.device("cuda:0"), T=32)
eng.keep_image = False
eng.set_schedule(torch.arange(32).reshape(1, 32).int())
eng.set_perceptual(Vgg16Features(layers_weights=[1, 1 / 16, 1 / 8, 1 / 4, 1], weights="random"), precision=prec)
for _ in range(6):
    eng.step(None, True, True)
torch.cuda.synchronize()
print("ok", eng.losses()["vgg"])
