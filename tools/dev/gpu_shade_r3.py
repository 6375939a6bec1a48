"""round-3 shader-backward experiments on the bench workload (C3: 32 frames, 512^2): time of harp_shade_bwd alone (single stream, HIP
events) for the face-staged kernel and the round-2 wave kernel, their ablations, gradient agreement between the two, and the step.
Environment: HARP_LIB_PATH (variant .so), HARP_SHADE_LDS_PAD (bytes of dynamic LDS: lowers occupancy), HARP_SHADE_BWD_OLD=1."""
import ctypes, os, sys, time
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

def timeit(flags=0, n=20):
    a.debug_skip = flags
    for _ in range(3): L.harp_shade_bwd(ctypes.byref(a), _lib.stream())
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): L.harp_shade_bwd(ctypes.byref(a), _lib.stream())
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n

tag = f"lib={os.path.basename(os.environ.get('HARP_LIB_PATH', 'default'))} pad={os.environ.get('HARP_SHADE_LDS_PAD', '0')} old={os.environ.get('HARP_SHADE_BWD_OLD', '0')} S={S} {KIND}"
sh = 8
cases = [("full", 0), ("notex", 1 << sh), ("nowin", 2 << sh), ("novtx", 4 << sh), ("noflush", 8 << sh),
         ("math only", (1 | 2 | 4 | 8) << sh), ("no pixels", 32 << sh), ("dispatch only", 64 << sh), ("first kernel (round 1)", 64)]
if os.environ.get("R3_QUICK"):
    cases = [c for c in cases if c[0] in ("full",)]
print(tag, "|", " | ".join(f"{k}={timeit(fl):.4f}" for k, fl in cases), flush=True)

if not os.environ.get("R3_NOCHECK"):
    # gradient agreement with the round-1 barrier-synchronised kernel (debug_skip = 64): one launch of each into cleared buffers
    def grads(flags):
        a.debug_skip = flags
        eng.gs_zero.zero_(); eng.gs_zero_late.zero_(); eng.gs_mesh.zero_()      # (g_vd / g_joints_m are a segment of their own since the third stream)
        L.harp_shade_bwd(ctypes.byref(a), _lib.stream()); torch.cuda.synchronize()
        return {k: eng.s[k].clone() for k in ("g_vd", "g_n2", "g_ndc_c", "g_zl", "g_light_pos", "g_colors", "g_light_R", "g_light_T", "g_nmap_n")} | \
               {"g_tex": eng.grads["texture"].clone(), "loss": eng.loss_vec.clone()}
    g_new, g_old = grads(0), grads(64)
    rel = lambda x, y: ((x.double() - y.double()).norm() / (y.double().norm() + 1e-30)).item()
    print("  wave kernel vs round-1 kernel, rel-L2:", {k: f"{rel(g_new[k], g_old[k]):.1e}" for k in g_new}, flush=True)
    eng._graphs = {}
    eng.auto_draw = True
    eng.set_schedule(torch.arange(32).reshape(1, 32).to(torch.int32))
    for _ in range(5): eng.step(None, True, True)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(100): eng.step(None, True, True)
    torch.cuda.synchronize(); print(f"  step={(time.perf_counter() - t) / 100 * 1e3:.4f} ms", flush=True)
