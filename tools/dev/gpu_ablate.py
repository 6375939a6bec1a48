import sys, time; sys.path.insert(0,'.')
import torch, ctypes
import bench
from harp_amd import _lib
eng, focal = bench.build_engine(0, 1, torch.device('cuda'), T=32, img=512, B=32)
fid = torch.arange(32)
eng.step(fid, True, True, use_graph=False); torch.cuda.synchronize()
L = _lib.lib()
a = eng._shade_struct(32, True)
def timeit(flags, n=10):
    a.debug_skip = flags
    for _ in range(2): L.harp_shade_bwd(ctypes.byref(a), _lib.stream())
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): L.harp_shade_bwd(ctypes.byref(a), _lib.stream())
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
cov = (eng.s['face_c']>=0).float().mean().item()
print('coverage', cov, 'active px', cov*32*512*512)
for name, fl in [('full',0),('no tex',1),('no nmap',2),('no tex+nmap',3),('no zl',4),('no vertex hash',8),('no tex/nmap/zl',7),('none of the scatters',15)]:
    print(f'{name:22s} {timeit(fl):8.3f} ms')
