"""workload model of the camera-view tile kernels on the bench scene (torch arithmetic on the engine's NDC vertices): staged faces per tile
(dilated boxes), faces with a pixel range in the hard pass (vertex boxes), 4x4 block visits of both, per frame and per tile"""
import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, bench
from harp_amd import ops
S = int(os.environ.get("R3_S", "512")); KIND = os.environ.get("R3_KIND", "hand")
eng, focal = bench.build_engine(0, 1, torch.device('cuda'), T=32, img=S, B=32, kind=KIND)
eng.keep_image = True
eng.step(torch.arange(32), True, True, use_graph=False); torch.cuda.synchronize()
ndc = eng.s["ndc_c"][:32].double(); faces = eng.topo.faces.long()
B, F = 32, faces.shape[0]
x = ndc[:, :, 0][:, faces]; y = ndc[:, :, 1][:, faces]                      # (B,F,3)
r = ops.SIL_BLUR ** 0.5
hs = S / 2
def rng(lo, hi, slack):                                                     # pixel-centre range of NDC interval [lo, hi]
    p0 = torch.ceil((1 - hi) * hs - 0.5 - slack).clamp(min=0); p1 = torch.floor((1 - lo) * hs - 0.5 + slack).clamp(max=S - 1)
    return p0.long(), p1.long()
out = {}
for name, rr in (("dilated", r), ("vertex", 0.0)):
    x0, x1 = rng(x.min(-1).values - rr, x.max(-1).values + rr, 0.05); y0, y1 = rng(y.min(-1).values - rr, y.max(-1).values + rr, 0.05)
    ok = (x0 <= x1) & (y0 <= y1)
    tx0, tx1, ty0, ty1 = x0 // 16, x1 // 16, y0 // 16, y1 // 16
    ntile = torch.where(ok, (tx1 - tx0 + 1) * (ty1 - ty0 + 1), 0)
    # blocks: 4x4 blocks relative to each tile origin == global 4x4 grid (16 is a multiple of 4) when ranges are clipped per tile: count
    # per tile sum of ((x1c-x0c)>>2)+1: not the global grid (the walk starts at x0, not at a multiple of 4): do it per tile
    nt = S // 16
    per_tile_faces = torch.zeros(B, nt, nt, dtype=torch.long, device=x.device)
    per_tile_blocks = torch.zeros(B, nt, nt, dtype=torch.long, device=x.device)
    for b in range(B):
        idx = ok[b].nonzero().flatten()
        for dx in range(int((tx1 - tx0).max()) + 1):
            for dy in range(int((ty1 - ty0).max()) + 1):
                tx, ty = tx0[b, idx] + dx, ty0[b, idx] + dy
                m = (tx <= tx1[b, idx]) & (ty <= ty1[b, idx])
                cx0 = torch.maximum(x0[b, idx], tx * 16); cx1 = torch.minimum(x1[b, idx], tx * 16 + 15)
                cy0 = torch.maximum(y0[b, idx], ty * 16); cy1 = torch.minimum(y1[b, idx], ty * 16 + 15)
                nb = (((cx1 - cx0) >> 2) + 1) * (((cy1 - cy0) >> 2) + 1)
                per_tile_faces[b].index_put_((ty[m], tx[m]), torch.ones_like(tx[m]), accumulate=True)
                per_tile_blocks[b].index_put_((ty[m], tx[m]), nb[m], accumulate=True)
    act = per_tile_faces > 0
    out[name] = (per_tile_faces, per_tile_blocks)
    print(name, "tiles with faces", int(act.sum()), "faces/tile mean %.1f max %d" % (per_tile_faces[act].float().mean(), per_tile_faces.max()),
          "(face,tile) pairs", int(per_tile_faces.sum()), "blocks/tile mean %.1f max %d" % (per_tile_blocks[act].float().mean(), per_tile_blocks.max()),
          "block visits", int(per_tile_blocks.sum()))
pf, pb = out["vertex"]
rounds = (pf + 15) // 16
print("scan rounds/tile (16 groups): mean %.2f max %d" % (rounds[pf > 0].float().mean(), rounds.max()))
cov = (eng.s["face_c"][:32] >= 0)
print("covered px", int(cov.sum()), "tiles with covered px", int((cov.view(32, S // 16, 16, S // 16, 16).sum((2, 4)) > 0).sum()))
