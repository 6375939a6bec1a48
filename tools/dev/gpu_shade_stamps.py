"""phase timing of the shader backward from inside the kernel (variant built with -DSHADE_STAMPS: tools/dev/build_variant.sh stamps
"-DSHADE_STAMPS" shade_bwd): average shader-clock cycles a working wave spends in each phase, alone on the GPU"""
import ctypes, os, sys
os.environ.setdefault("HARP_LIB_PATH", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "harp_amd", "csrc", "variants", "libharp_stamps.so"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch, bench
from harp_amd import _lib
S = int(os.environ.get("R3_S", "512")); KIND = os.environ.get("R3_KIND", "hand")
eng, focal = bench.build_engine(0, 1, torch.device("cuda"), T=32, img=S, B=32, kind=KIND)
eng.keep_image = False
fid = torch.arange(32)
eng.auto_draw = False
eng.draw_texture_offsets()
eng.fid.copy_(fid.int().cuda()); eng.tfid.copy_(fid.int().cuda())
eng.set_stage(True, True)
eng.forward_backward(True, True); torch.cuda.synchronize()
L, p = _lib.lib(), _lib.ptr
a = eng._shade_struct(32, True)
a.l1_target, a.l1_mask, a.l1_fid = p(eng.y_true), p(eng.y_sil_col), p(eng.tfid)
a.l1_w, a.l1_loss, a.l1_grad = eng.w_vec.data_ptr() + 24, eng.loss_vec.data_ptr() + 24, p(eng.s["g_rgb"])
a.g_rgb = None
raw = ctypes.CDLL(os.environ["HARP_LIB_PATH"])
buf = (ctypes.c_ulonglong * 16)()
raw.harp_debug_shade_stamps(buf)                       # reset
N = 10
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(N): L.harp_shade_bwd(ctypes.byref(a), _lib.stream())
e1.record(); torch.cuda.synchronize()
raw.harp_debug_shade_stamps(buf)
n = buf[0]
names = ["", "decode+compaction+clears", "face+vertex loads -> uv", "texel fetch + tangent frame", "lighting + shadow taps", "colour + backward math + reloads",
         "17 wave sums", "merge + vertex table", "shadow window + flush", "texel table + flush", "vertex flush + drain"]
tot = sum(buf[k] for k in range(1, 11))
print("waves with active pixels per launch", n // N, "kernel ms", e0.elapsed_time(e1) / N, "mean wave life cycles", tot / n)
for k in range(1, 11):
    print("  %-36s %8.0f cycles  %5.1f %%" % (names[k], buf[k] / n, 100.0 * buf[k] / tot))
