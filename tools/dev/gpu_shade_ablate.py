"""ablation timing of harp_shade_bwd (production wave kernel: bits 8+, first kernel: bit 6 + bits 0-5), single stream, bench workload"""
import sys, os, time, ctypes; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, bench
from harp_amd import _lib
eng, focal = bench.build_engine(0, 1, torch.device('cuda'), T=32, img=512, B=32)
eng.keep_image = False
fid = torch.arange(32)
eng.step(fid, True, True, use_graph=False); torch.cuda.synchronize()
L = _lib.lib()
a = eng._shade_struct(32, True)
p = _lib.ptr
a.l1_target, a.l1_mask, a.l1_fid = p(eng.y_true), p(eng.y_sil_col), p(eng.tfid)
a.l1_w, a.l1_loss, a.l1_grad = eng.w_vec.data_ptr() + 24, eng.loss_vec.data_ptr() + 24, p(eng.s["g_rgb"])
a.g_rgb = None
def timeit(flags=0, n=20):
    a.debug_skip = flags
    for _ in range(3): L.harp_shade_bwd(ctypes.byref(a), _lib.stream())
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): L.harp_shade_bwd(ctypes.byref(a), _lib.stream())
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
cases = [("wave full", 0), ("wave notex", 256), ("wave nowin", 512), ("wave novtx", 1024), ("wave noflush", 2048), ("wave math only", 256 | 512 | 1024 | 2048),
         ("wave nothing", 8192), ("wave dispatch only", 16384), ("first kernel", 64), ("first none", 64 | 15 | 48)]
print(" | ".join(f"{k}={timeit(fl):.3f}" for k, fl in cases))
