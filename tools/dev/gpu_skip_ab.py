"""TIMING ONLY (results are wrong): what would the replayed step cost without some of its small kernels?  Library entry points named in a
configuration return HARP_OK without launching; python tools/dev/gpu_skip_ab.py"""
import sys, os, time, types; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, bench
from harp_amd import _lib
real = _lib.lib()
skip = set()
class Proxy:
    def __getattr__(self, name):
        f = getattr(real, name)
        if name in skip:
            return lambda *a, **k: 0
        return f
_lib.lib = lambda: Proxy()
import harp_amd.engine as E
eng, focal = bench.build_engine(0, 1, torch.device('cuda'))
eng.keep_image = False
eng.set_schedule(torch.arange(256).reshape(-1, 32).int())
gs_mesh = eng.gs_mesh
eng.set_lr(0.0, 0.0)          # parameters never move: skipped clears cannot change the geometry that is timed
sched_next = eng._schedule_next
CONFIGS = {
    "base": (),
    "no schedule kernel": ("SCHED",),
    "+ no adam tick": ("SCHED", "harp_adam_tick"),
    "+ normalize3 / close_z / 2nd smooth / sumsq gone (fused elsewhere)": ("SCHED", "harp_adam_tick", "harp_normalize3_fwd", "harp_close_to_z_reg", "harp_sum_squares", "SMOOTH2"),
    "+ no third-stream fill, no kps": ("SCHED", "harp_adam_tick", "harp_normalize3_fwd", "harp_close_to_z_reg", "harp_sum_squares", "SMOOTH2", "MESHFILL", "harp_kps_loss"),
    "+ no normalize3_bwd": ("SCHED", "harp_adam_tick", "harp_normalize3_fwd", "harp_close_to_z_reg", "harp_sum_squares", "SMOOTH2", "MESHFILL", "harp_kps_loss", "harp_normalize3_bwd"),
}
def apply(cfg):
    skip.clear(); skip.update(c for c in cfg if c.startswith("harp_"))
    eng.gs_mesh = types.SimpleNamespace(zero_=lambda: None) if "MESHFILL" in cfg else gs_mesh
    if "SCHED" in cfg:
        eng._schedule_next = lambda: setattr(eng, "_loss_cleared", True)
    else:
        eng._schedule_next = sched_next
    if "SMOOTH2" in cfg:
        n = [0]
        def smooth(*a):
            n[0] += 1
            return 0 if n[0] % 2 == 0 else real.harp_texture_smooth_reg(*a)
        Proxy.harp_texture_smooth_reg = property(lambda self: smooth)
    elif hasattr(Proxy, "harp_texture_smooth_reg"):
        del Proxy.harp_texture_smooth_reg
    eng._graphs = {}
def run(n=60):
    for _ in range(6): eng.step(None, True, True)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): eng.step(None, True, True)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
for rep in range(3):
    for name, cfg in CONFIGS.items():
        apply(cfg)
        print("%.4f ms/step  %s" % (run(), name), flush=True)
