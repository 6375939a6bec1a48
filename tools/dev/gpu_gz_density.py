"""how much of the light view does the shadow-map gradient image touch?  tiles (16x16) with a non-zero g_zl entry against the tiles the depth
backward visits (all tiles of super-tiles that hold a light-view face): python tools/dev/gpu_gz_density.py [arm 1024]"""
import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, bench
kind = sys.argv[1] if len(sys.argv) > 1 else "hand"
img = int(sys.argv[2]) if len(sys.argv) > 2 else 512
eng, focal = bench.build_engine(0, 1, torch.device('cuda'), T=32, img=img, kind=kind)
eng.keep_image = False
eng.consume_gzl = False; eng.keep_depth = False
eng.set_stage(True, True)
eng.fid.copy_(torch.arange(32, dtype=torch.int32)); eng.tfid.copy_(torch.arange(32, dtype=torch.int32))
eng.forward_backward(True, True)
torch.cuda.synchronize()
S, B = img, 32
gz = eng.s["g_zl"].view(B, S // 16, 16, S // 16, 16)
tile_nz = (gz != 0).any(dim=4).any(dim=2)                       # (B, S/16, S/16)
face = eng.s["face_l"].view(B, S // 64, 64, S // 64, 64)
st_has = (face >= 0).any(dim=4).any(dim=2)                      # super-tiles with a covered pixel (~ super-tiles that hold a face)
visited = st_has.repeat_interleave(4, 1).repeat_interleave(4, 2)
tface = (eng.s["face_l"].view(B, S // 16, 16, S // 16, 16) >= 0).any(dim=4).any(dim=2)
print(kind, img, "tiles visited %d, tiles with a face pixel %d, tiles with a gradient %d (%.1f %% of visited); pixels with gradient %.1f %% of visited pixels" % (
    visited.sum().item(), tface.sum().item(), tile_nz.sum().item(), 100.0 * tile_nz.sum().item() / visited.sum().item(),
    100.0 * (eng.s["g_zl"] != 0).sum().item() / (visited.sum().item() * 256)))
