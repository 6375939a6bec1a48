import sys, ctypes; sys.path.insert(0, '.')
import torch, bench
from harp_amd import _lib
eng, focal = bench.build_engine(0, 1, torch.device('cuda'), T=32, img=512, B=32)
eng.step(torch.arange(32), True, True, use_graph=False); torch.cuda.synchronize()
L = _lib.lib()
a = eng._shade_struct(32, True)
out = {}
for flag in (64, 0):
    a.debug_skip = flag
    eng.s['g_zl'].zero_()
    L.harp_shade_bwd(ctypes.byref(a), _lib.stream()); torch.cuda.synchronize()
    out[flag] = eng.s['g_zl'].clone()
for fl in ():
    dd = (out[fl] - out[64]).abs(); print('flag', fl, 'max abs diff', dd.max().item(), 'sum', out[fl].double().sum().item())
d = (out[0] - out[64]).abs()
print('g_zl sum', out[64].double().sum().item(), out[0].double().sum().item(), 'abs sum', out[64].abs().double().sum().item(), out[0].abs().double().sum().item())
print('max abs diff', d.max().item(), 'ref max', out[64].abs().max().item(), 'n diff > 1e-6*max', (d > 1e-6 * out[64].abs().max()).sum().item())
idx = (d > 1e-6 * out[64].abs().max()).nonzero()[:10]
print(idx)
for i in idx[:0]:
    b, y, x = i.tolist(); print(out[64][b, y-1:y+2, x-2:x+3], out[0][b, y-1:y+2, x-2:x+3])
