import sys, time; sys.path.insert(0,'.')
import torch
import bench
from harp_amd import ops, _lib
eng, focal = bench.build_engine(0, 1, torch.device('cuda'), T=32, img=512, B=32)
eng.step(torch.arange(32), True, True, use_graph=False); torch.cuda.synchronize()
nd = eng.s['ndc_c'].clone(); faces = eng.topo.faces; ws = eng.s['ws_c']
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
for name, x in (('normal', nd), ('all culled (z<0)', nd*torch.tensor([1,1,-1.],device='cuda')), ('tiny (whole hand in 1 tile)', torch.cat([nd[...,:2]*0.02, nd[...,2:]],-1).contiguous())):
    for soft in (True, False):
        us = t(lambda: ops.rasterize_fwd(x, faces, 512, soft=soft, blur_radius=ops.SIL_BLUR, sigma=ops.SIL_SIGMA, ws=ws))
        print(f'{name:30s} soft={soft}: {us:8.1f} us')
# coverage stats
f,_,_,_ = ops.rasterize_fwd(nd, faces, 512, soft=False, ws=ws)
cov = (f>=0).float().mean().item(); print('coverage', cov)
import numpy as np
nsx=8; F=faces.shape[0]
n_recs = 32*F*64; n_bbs = 32*F*16
cnt = ws[n_recs+n_bbs+32*64*F*4: n_recs+n_bbs+32*64*F*4 + 32*64*4].view(torch.int32).cpu().numpy().reshape(32,64)
print('super-tile list len: mean nonzero', cnt[cnt>0].mean(), 'max', cnt.max(), 'nonzero frac', (cnt>0).mean(), 'sum per frame', cnt.sum(1).mean(), 'F', F)
bb = ws[n_recs:n_recs+n_bbs].view(torch.float32).view(32,F,4)
ok = bb[...,0] < bb[...,1]
S=512
def pix(ndc): return (S*(1-ndc)-1)/2     # ndc -> pixel index (float)
x0 = pix(bb[...,1]).clamp(0,S-1); x1 = pix(bb[...,0]).clamp(0,S-1); y0 = pix(bb[...,3]).clamp(0,S-1); y1 = pix(bb[...,2]).clamp(0,S-1)
vis = ok & (x1>=0) & (y1>=0)
w = (x1-x0)[vis]; h=(y1-y0)[vis]
print('faces kept', vis.float().mean().item(), 'bbox w mean', w.mean().item(), 'h mean', h.mean().item(), 'area mean', (w*h).mean().item())
nx = (torch.floor(x1/16)-torch.floor(x0/16)+1)[vis]; ny4 = (torch.floor(y1/4)-torch.floor(y0/4)+1)[vis]; ny16=(torch.floor(y1/16)-torch.floor(y0/16)+1)[vis]
print('strip pairs per frame', (nx*ny4).sum().item()/32, 'tile pairs per frame', (nx*ny16).sum().item()/32, 'total strip pairs', (nx*ny4).sum().item())
