"""event-timed harp_mesh_kps_terms of the bench workload (single stream, eager) for the library in HARP_LIB_PATH"""
import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, bench
kind, img = os.environ.get("HARP_TL_KIND", "hand"), int(os.environ.get("HARP_TL_IMG", "512"))
eng, focal = bench.build_engine(0, 1, torch.device('cuda'), kind=kind, img=img, T=32 if kind == "arm" else 256)
eng.keep_image = False
eng.set_schedule(torch.arange(32).reshape(1, 32).int())
for _ in range(3): eng.step(None, True, True, use_graph=False)
from harp_amd import _lib
L, p, s, tp = _lib.lib(), _lib.ptr, eng.s, eng.topo
lane = eng._lane
w = lane["w_vec"]; lv = eng.loss_acc
def call():
    eng._ck(L.harp_mesh_kps_terms(p(s["vd"]), p(eng.ref_verts), p(tp.nbr_off), p(tp.nbr_idx), p(tp.nc_pairs), p(tp.vp_off), p(tp.vp_idx), eng.B, tp.V,
                                  tp.nc_pairs.shape[0], tp.E, w.data_ptr() + 12, lv.data_ptr() + 12, p(s["g_vd"]), p(eng.init_joints), p(eng.fid), p(s["joints_m"]),
                                  eng.n_joints, w.data_ptr() + 4, lv.data_ptr() + 4, p(s["g_joints_m"]), _lib.stream()), "mesh_kps")
for _ in range(5): call()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50): call()
b.record(); torch.cuda.synchronize()
print(os.path.basename(os.environ.get("HARP_LIB_PATH", "default")), kind, "mesh_kps_terms %.2f us" % (a.elapsed_time(b) / 50 * 1e3))
