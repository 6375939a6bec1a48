"""harp_texture_terms alone on the bench engine's maps: ms per launch (HARP_TEXTERMS_LDS: the LDS fence; HARP_TT_DBG ablations live in tools/dev/variants/texture_terms_timing_ablations.patch)"""
import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, bench
from harp_amd import _lib
eng, focal = bench.build_engine(0, 1, torch.device('cuda'), T=32, B=32)
eng.keep_image = False
eng.step(torch.arange(32), True, True, use_graph=False); torch.cuda.synchronize()
L, p, V = _lib.lib(), _lib.ptr, eng.topo.V
w, l = eng.w_vec, torch.zeros(16, device="cuda")
wp = lambda i: w.data_ptr() + 4 * i
lp = lambda i: l.data_ptr() + 4 * i
def run(n):
    for _ in range(n):
        L.harp_texture_terms(p(eng.params["texture"]), p(eng.params["normal_map"]), p(eng.uv_mask), p(eng.dist_albedo), p(eng.dist_normal), eng.Ht, eng.Wt, 0.2,
                             wp(7), lp(7), p(eng.grads["texture"]), wp(8), lp(8), p(eng.grads["normal_map"]), p(eng.params["verts_disps"]), V, wp(2), lp(2),
                             p(eng.grads["verts_disps"]), None, _lib.stream())
run(3); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(20); e1.record(); torch.cuda.synchronize()
print("HARP_TT_DBG", os.environ.get("HARP_TT_DBG", "0"), "HARP_TEXTERMS_LDS", os.environ.get("HARP_TEXTERMS_LDS", "default"), "texture_terms %.4f ms" % (e0.elapsed_time(e1) / 20), "mask share %.3f" % (eng.uv_mask != 0).float().mean().item())
