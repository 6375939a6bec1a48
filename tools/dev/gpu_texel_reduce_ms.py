"""harp_texel_reduce alone on the records of one bench step (C3 by default; `arm` = C5's share): ms per launch.  HARP_TREC_DBG selects
the ablations of tools/dev/variants/texel_reduce_timing_ablations.patch (build_variant.sh NAME "" texel_reduce <patch>, HARP_LIB_PATH; bit 4 = counters kept is forced here so that every launch sees the same lists — without the patch the first launch consumes them)."""
import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
os.environ["HARP_TREC_DBG"] = str(int(os.environ.get("HARP_TREC_DBG", "0")) | 4)
import torch, bench
from harp_amd import _lib
kind = sys.argv[1] if len(sys.argv) > 1 else "hand"
img = 1024 if kind == "arm" else 512
eng, focal = bench.build_engine(0, 1, torch.device('cuda'), T=32, img=img, B=32, kind=kind)
eng.keep_image = False
eng.step(torch.arange(32), True, True, use_graph=False); torch.cuda.synchronize()
rec, cnt, cap = eng._texel_record_buffers()
c = cnt[::16][:-1]
print("records", int(c.sum()), "bins used", int((c > 0).sum()), "max", int(c.max()), "cap", cap, "chunks", int(((c + 2047) // 2048).sum()))
L, p = _lib.lib(), _lib.ptr
gt, gn = (torch.zeros(eng.Ht * eng.Wt * 3, dtype=torch.float64, device="cuda") for _ in range(2))
def run(n):
    for _ in range(n): L.harp_texel_reduce(p(rec), p(cnt), cap, eng.Ht, eng.Wt, p(gt), p(gn), int(c.sum()), _lib.stream())
run(3); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(20); e1.record(); torch.cuda.synchronize()
print("HARP_TREC_DBG", os.environ["HARP_TREC_DBG"], "texel_reduce %.4f ms" % (e0.elapsed_time(e1) / 20))
# duplicates: how many of 64 consecutive records of a list share their top-left texel
if os.environ.get("TREC_STATS"):
    import numpy as np
    cc = c.cpu().numpy(); tot = 0; dup = 0; mx = []
    for b in np.nonzero(cc)[0][:40]:
        keys = rec[(b * 9) * cap:(b * 9) * cap + int(cc[b])].view(torch.int32).cpu().numpy()
        for i in range(0, len(keys) - 63, 64 * 7):
            u, n = np.unique(keys[i:i + 64], return_counts=True); tot += 64; dup += 64 - len(u); mx.append(n.max())
    print("sampled groups of 64 consecutive records: distinct share %.3f, max lanes per texel mean %.1f p90 %d max %d" % (1 - dup / tot, np.mean(mx), np.percentile(mx, 90), np.max(mx)))
