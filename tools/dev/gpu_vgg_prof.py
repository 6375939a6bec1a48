"""Where the perceptual term's time goes (B=32, 512^2): event-timed pieces of Engine._perceptual_term, argv[1] = fp32|bf16.
(rocprofv3 over MIOpen's first-call kernel search takes tens of minutes: do not profile this script with it.)"""
import sys
import torch
sys.path.insert(0, ".")
import bench
from harp_amd.model.vgg import Vgg16Features
T = 32
eng, _ = bench.build_engine(0, 1, torch.device("cuda:0"), T=T)
vgg = Vgg16Features(layers_weights=[1, 1 / 16, 1 / 8, 1 / 4, 1], weights="random").cuda()
ac = torch.bfloat16 if sys.argv[1] == "bf16" else None
eng.set_perceptual(vgg, autocast=ac)


def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        out = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, out


idx = torch.arange(32, device="cuda")
m = eng.y_sil_col[idx].unsqueeze(-1)
leaf = eng.y_true[idx].clone().mul_(0.9).requires_grad_(True)
x = (leaf * m).permute(0, 3, 1, 2)
with torch.no_grad():
    t, _ = timed(lambda: eng._vgg_run(x.detach()))
print("features, no grad        %7.2f ms" % t)
t, fp = timed(lambda: eng._vgg_run(x))
print("features, grad recorded  %7.2f ms" % t)
t, ft = timed(lambda: [x.detach().flatten(start_dim=1)] + [c[idx] for c in eng._vgg_cache])
print("target gather            %7.2f ms" % t)
n = sum(f.shape[1] for f in fp) * 32
t, loss = timed(lambda: sum(w * (a - b).abs().sum() for w, a, b in zip(vgg.layers_weights, fp, ft)) / n)
print("L1 over features         %7.2f ms" % t)
t, _ = timed(lambda: torch.autograd.grad(loss, leaf, retain_graph=True))
print("backward to the image    %7.2f ms" % t)
for nm, sl in (("slice1", vgg.slice1), ("slice2", vgg.slice2), ("slice3", vgg.slice3), ("slice4", vgg.slice4)):
    with torch.no_grad():
        h = x.detach()
        for k in range(1, int(nm[-1])):
            h = getattr(vgg, f"slice{k}")(h)
        if ac is not None:
            with torch.autocast("cuda", dtype=ac):
                t, _ = timed(lambda: sl(h))
        else:
            t, _ = timed(lambda: sl(h))
    print("%s forward           %7.2f ms" % (nm, t))
