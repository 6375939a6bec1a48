import sys, time, ctypes; sys.path.insert(0,'.')
import torch, bench
from harp_amd import _lib, ops
eng, focal = bench.build_engine(0, 1, torch.device('cuda'), T=32, img=512, B=32)
eng.micro = 1
eng.step(torch.arange(32), True, True, use_graph=False); torch.cuda.synchronize()
L, p, s, tp = _lib.lib(), _lib.ptr, eng.s, eng.topo
a = eng._shade_struct(32, True)
B,V,F,S = 32, tp.V, tp.F, 512
ws2 = ops.rasterize_workspace(B, F, S, 'cuda'); f2 = torch.empty_like(s['face_c']); a2 = torch.empty_like(s['alpha'])
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def raster(st): L.harp_rasterize_fwd(p(s['ndc_c']), p(tp.faces), B, V, F, S, 1, ops.SIL_BLUR, ops.SIL_SIGMA, p(ws2), p(f2), None, p(a2), st.cuda_stream)
def shbwd(st): L.harp_shade_bwd(ctypes.byref(a), st.cuda_stream)
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
print('raster alone', timeit(lambda: raster(s1)))
print('shade_bwd alone', timeit(lambda: shbwd(s1)))
print('sequential', timeit(lambda: (raster(s1), shbwd(s1))))
print('concurrent', timeit(lambda: (raster(s1), shbwd(s2))))
