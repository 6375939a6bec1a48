"""event-time the hand layer forward / backward launches alone, after evicting L2 with a large copy (the state they see inside a step)"""
import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, ctypes
from harp_amd import synth, _lib
from harp_amd.manopth.manolayer import ManoDeviceModel
tpl = synth.load_template("hand"); model = synth.make_mano_model(tpl, seed=0)
dm = ManoDeviceModel(model, torch.device("cuda"))
B = 32
L, p = _lib.lib(), _lib.ptr
pose, betas, trans = torch.randn(B, 48, device="cuda") * 0.2, torch.randn(B, 10, device="cuda"), torch.randn(B, 3, device="cuda") * 0.01
ws = torch.empty(L.harp_lbs_mano_ws_floats(B), device="cuda")
v, j = torch.empty(B, 778, 3, device="cuda"), torch.empty(B, 21, 3, device="cuda")
gv, gj = torch.randn(B, 778, 3, device="cuda"), torch.randn(B, 21, 3, device="cuda")
gp, gb, gt = torch.empty(B, 48, device="cuda"), torch.empty(B, 10, device="cuda"), torch.empty(B, 3, device="cuda")
big = torch.empty(64 << 20, device="cuda"); big2 = torch.empty_like(big)
def run(fn, n=20):
    tot = 0.0
    for _ in range(n):
        big2.copy_(big)                                   # 512 MB through L2 / MALL
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / n * 1e3
fwd = lambda: L.harp_lbs_mano_fwd(ctypes.byref(dm.struct), p(pose), p(betas), p(trans), B, p(ws), p(v), p(j), _lib.stream())
bwd = lambda: L.harp_lbs_mano_bwd(ctypes.byref(dm.struct), p(pose), p(betas), p(trans), B, p(ws), p(gv.clone()), p(gj), p(gp), p(gb), p(gt), _lib.stream())
fwd(); bwd(); torch.cuda.synchronize()
print("lbs fwd %.1f us   lbs bwd %.1f us (cold caches, incl. a clone in bwd)" % (run(fwd), run(bwd)))
