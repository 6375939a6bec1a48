"""run the three raster launches a few times (for PMC passes)"""
import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, bench, ctypes
from harp_amd import ops, _lib
eng, focal = bench.build_engine(0, 1, torch.device('cuda'), T=32, img=512, B=32)
eng.overlap = False
for _ in range(4):
    eng.step(torch.arange(32), True, True, use_graph=False)
torch.cuda.synchronize()
