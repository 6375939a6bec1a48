"""bbox statistics of the eroded photometric masks of the bench scene (what an exactly-sparse VGG term could crop to)"""
import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, bench
eng, focal = bench.build_engine(0, 1, torch.device('cuda'), T=32, B=32)
m = eng.y_sil_col > 0
ys = m.any(2); xs = m.any(1)
S = m.shape[1]
idx = torch.arange(S, device=m.device)
y0 = torch.where(ys, idx, S).min(1).values; y1 = torch.where(ys, idx, -1).max(1).values
x0 = torch.where(xs, idx, S).min(1).values; x1 = torch.where(xs, idx, -1).max(1).values
h, w = (y1 - y0 + 1).float(), (x1 - x0 + 1).float()
print("mask bbox h: mean %.0f max %.0f  w: mean %.0f max %.0f  covered frac %.3f" % (h.mean(), h.max(), w.mean(), w.max(), m.float().mean()))
for halo in (48, 96):
    H = ((h.max() + 2 * halo + 7) // 8 * 8).clamp(max=S); W = ((w.max() + 2 * halo + 7) // 8 * 8).clamp(max=S)
    print("halo", halo, "common crop", int(H), "x", int(W), "=", float(H * W) / S / S, "of the image")
