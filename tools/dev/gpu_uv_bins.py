"""How the active pixels of a step spread over 32x32-texel UV tiles (the owner bins of the texel reduce, csrc/texel_reduce.hip):
records per bin, distinct texels touched, how many bins a 64-pixel wave of the compacted order feeds."""
import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, bench
from harp_amd import ops
kind = sys.argv[1] if len(sys.argv) > 1 else "hand"
img = 1024 if kind == "arm" else 512
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
eng, focal = bench.build_engine(0, 1, torch.device('cuda'), T=B, img=img, B=B, kind=kind)
eng.keep_image = True
eng.step(torch.arange(B), True, True, use_graph=False); torch.cuda.synchronize()
S = eng.S
f = eng.s["face_c"][:B]; m = eng.y_sil_col[:B]
act = (f >= 0) & (m != 0)
p2f, z, bary, d = ops.rasterize_fragments(eng.s["ndc_c"][:B].contiguous(), eng.topo.faces, S, 0.0, 1, packed=False)
p2f = p2f[..., 0]; bary = bary[..., 0, :]
fu = eng.topo.faces_uvs.long(); vu = eng.topo.verts_uvs
ff = p2f.clamp(min=0).long()
uv = (vu[fu[ff]] * bary.unsqueeze(-1)).sum(-2)          # (B,S,S,2)
Wt = eng.Wt
x = (uv[..., 0] * (Wt - 1)).clamp(0, Wt - 1); y = ((1 - uv[..., 1]) * (Wt - 1)).clamp(0, Wt - 1)
x0 = x.floor().long(); y0 = y.floor().long()
for tile in (32, 16):
    nb = (Wt + tile - 1) // tile
    bins = ((y0 // tile) * nb + x0 // tile)[act]
    cnt = torch.bincount(bins, minlength=nb * nb)
    nz = cnt[cnt > 0]
    print(f"tile {tile}: active px {int(act.sum())}, bins used {nz.numel()} of {nb*nb}, records per used bin mean {nz.float().mean():.0f} max {int(nz.max())} "
          f"p50 {int(nz.float().median())} p90 {int(nz.float().quantile(0.9))}; chunks of 4096: {int(((nz + 4095)//4096).sum())}, of 2048: {int(((nz + 2047)//2048).sum())}")
# bins per wave of the compacted order (16x16 tile, row-major compaction, 64 per wave)
nb = Wt // 32
bin32 = ((y0 // 32) * nb + x0 // 32)
nt = S // 16
a = act.view(B, nt, 16, nt, 16).permute(0, 1, 3, 2, 4).reshape(-1, 256)
bb = bin32.view(B, nt, 16, nt, 16).permute(0, 1, 3, 2, 4).reshape(-1, 256)
sel = a.sum(1) > 0
a, bb = a[sel].cpu(), bb[sel].cpu()
import collections
hist = collections.Counter(); waves = 0
for i in range(0, a.shape[0], max(1, a.shape[0] // 3000)):
    v = bb[i][a[i]]
    for w0 in range(0, v.numel(), 64):
        hist[int(v[w0:w0 + 64].unique().numel())] += 1; waves += 1
print("distinct 32x32 bins per wave (sampled):", sorted(hist.items()), "waves", waves)
key = (y0 * Wt + x0)[act]
print("distinct top-left texels", key.unique().numel(), "of", int(act.sum()), "records")
