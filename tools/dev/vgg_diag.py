"""diagnostic: float32 vs bf16-split VGG term on the bounded-mode test's inputs — feature differences per tap, gradient error vs torch float64"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from harp_amd.model.vgg import Vgg16Features
from harp_amd.model.vgg_hip import Vgg16Hip, active_tiles
DEV = "cuda:0"
S, N, T = 128, 3, 4
LW = [1, 1 / 16, 1 / 8, 1 / 4, 1]
vgg = Vgg16Features(layers_weights=LW, weights="random", seed=3)
g = torch.Generator().manual_seed(21)
rgb = torch.rand(N, S, S, 3, generator=g, dtype=torch.float64).requires_grad_(True)
y_true = torch.rand(T, S, S, 3, generator=g, dtype=torch.float64)
mask = torch.zeros(T, S, S, dtype=torch.float64)
mask[0, 70:100, 20:55] = 1.0
mask[1, 0:40, 90:128] = (torch.rand(40, 38, generator=g) > 0.2).double()
mask[3, 60:62, 60:62] = 0.5
rows = torch.tensor([1, 0, 3])
vgg64 = Vgg16Features(layers_weights=LW, weights=vgg.state_dict()).double()
m = mask[rows].unsqueeze(-1)
fp = vgg64.features((rgb * m).permute(0, 3, 1, 2), weighted=False)
ft = vgg64.features((y_true[rows] * m).permute(0, 3, 1, 2), weighted=False)
n = sum(f.shape[1] for f in fp) * N
for li in range(5):
    lw = [0.0] * 5; lw[li] = LW[li]
    want = sum(abs(w) * (a - b).abs().sum() for w, a, b in zip(lw, fp, ft)) / n
    (g_want,) = torch.autograd.grad(want, rgb, retain_graph=True)
    for prec in (0, 1):
        v = Vgg16Features(layers_weights=lw, weights=vgg.state_dict())
        hip = Vgg16Hip(v, DEV, prec)
        rgb_d, yt_d, mask_d = (t.detach().float().to(DEV).contiguous() for t in (rgb, y_true, mask))
        cache = hip.features(yt_d, mask_d, all_slots=True)
        g_rgb = torch.zeros(N, S, S, 3, device=DEV); loss = torch.zeros(1, device=DEV)
        hip.term(rgb_d, yt_d, mask_d, rows.int().to(DEV), cache, 1, g_rgb, loss)
        torch.cuda.synchronize()
        e = (g_rgb.double().cpu() - g_want)
        per = [(e[i].norm() / g_want[i].norm().clamp_min(1e-30)).item() for i in range(N)]
        print(f"row {li} prec {prec}: loss {loss.item():.3e} vs {want.item():.3e}; grad rel-L2 {(e.norm() / g_want.norm()).item():.2e}; per image {['%.1e' % p for p in per]}")
h0, h1 = Vgg16Hip(vgg, DEV, 0), Vgg16Hip(vgg, DEV, 1)
x = torch.rand(2, S, S, 3, device=DEV); mk = torch.ones(2, S, S, device=DEV)
f0, f1 = h0.features(x, mk), h1.features(x, mk)
print("tap feature rel diff float32 vs bf16 split:", [((a - b).norm() / a.norm()).item() for a, b in zip(f0, f1)])
