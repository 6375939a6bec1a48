import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, bench
eng, focal = bench.build_engine(0, 1, torch.device('cuda'), T=32, img=512, B=32)
eng.keep_image = True
fid = torch.arange(32)
eng.step(fid, True, True, use_graph=False); torch.cuda.synchronize()
f = eng.s["face_c"]; m = eng.y_sil_col[:32]
cov = f >= 0; act = cov & (m != 0)
print("nact", __import__("harp_amd.ops", fromlist=["x"]).rasterize_ws_nact(eng.s["ws_c"], eng.B, eng.topo.F, eng.S), "of", 32 * 64)
print("covered px", cov.sum().item(), "active px", act.sum().item(), "of", f.numel())
t = act.view(32, 32, 16, 32, 16).permute(0, 1, 3, 2, 4).reshape(32, 32, 32, 256)     # tiles
per_tile = t.sum(-1)
print("tiles with active", (per_tile > 0).sum().item(), "tiles with covered", (cov.view(32, 32, 16, 32, 16).permute(0, 1, 3, 2, 4).reshape(32,32,32,256).sum(-1) > 0).sum().item())
s = act.view(32, 128, 4, 32, 16).permute(0, 1, 3, 2, 4).reshape(32, 128, 32, 64).sum(-1)   # strips 16x4
print("strips with active", (s > 0).sum().item(), "mean active lanes in nonempty strips", s[s > 0].float().mean().item())
chunks = ((per_tile + 63) // 64).sum().item()
print("chunks after per-tile compaction", chunks, "hist per tile", torch.bincount(((per_tile + 63) // 64).flatten()).tolist())
