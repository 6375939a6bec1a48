"""time harp_shade_bwd (bucketed and direct flush) + the graph-replayed step for the library in HARP_LIB_PATH"""
import sys, os, time, ctypes; sys.path.insert(0, '.')
import torch, bench
from harp_amd import _lib
eng, focal = bench.build_engine(0, 1, torch.device('cuda'), T=32, img=512, B=32)
fid = torch.arange(32)
eng.step(fid, True, True, use_graph=False); torch.cuda.synchronize()
L = _lib.lib()
def timeit(a, flags=0, n=20):
    a.debug_skip = flags
    for _ in range(3): L.harp_shade_bwd(ctypes.byref(a), _lib.stream())
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): L.harp_shade_bwd(ctypes.byref(a), _lib.stream())
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
out = [os.path.basename(os.environ.get('HARP_LIB_PATH', 'default'))]
for mode in (False,):
    a = eng._shade_struct(32, True)
    out.append(' '.join(f'{k}={timeit(a, fl):.3f}' for k, fl in (('full', 0), ('notex', 3), ('novtx', 8), ('none', 15), ('notexflush', 16), ('novtxflush', 32), ('noflush', 48), ('noflush_nozl', 52), ('none', 15 | 48), ('none_noshadow', 15 | 48 | 64), ('none_nonmap', 15 | 48 | 128), ('none_noshadow_nonmap', 15 | 48 | 192))))
    eng._graphs = {}
    for _ in range(3): eng.step(fid, True, True, use_graph=True)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(50): eng.step(fid, True, True, use_graph=True)
    torch.cuda.synchronize(); out.append(f'step={(time.perf_counter() - t) / 50 * 1e3:.3f}ms')
print(' | '.join(out))
