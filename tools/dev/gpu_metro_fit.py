"""Wall time of the MANO-to-METRO vertex fit (500 + 700 Adam iterations, hand_utils.py:16-131) for B frames at once:
native loop replayed as a hipGraph / the same launches eagerly / torch autograd + torch.optim.Adam through the HIP hand layer."""
import sys, time
import torch
sys.path.insert(0, ".")
from harp_amd import synth
from harp_amd.manopth.manolayer import ManoLayer
from harp_amd.metro_modifications import hand_utils as hu

tpl = synth.load_template("hand")
model = synth.make_mano_model(tpl, seed=0)
layer = ManoLayer(flat_hand_mean=False, use_pca=False, model=model, device="cuda")
for B in (1, 64, 1024):
    g = torch.Generator().manual_seed(B)
    with torch.no_grad():
        verts, _ = layer(torch.cat((torch.randn(B, 3, generator=g) * 0.4, torch.randn(B, 45, generator=g) * 0.25), 1).cuda(),
                         (torch.randn(B, 10, generator=g) * 0.5).cuda(), (torch.randn(B, 3, generator=g) * 0.05 + torch.tensor([0, 0, 0.6])).cuda())
    pred = verts.cpu() / 1000.0
    res = {}
    for name, kw in (("hipGraph", dict(use_graph=True)), ("eager", dict(use_graph=False))):
        hu.optimize_for_mano_param(pred, layer, **kw)                       # warm
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = hu.optimize_for_mano_param(pred, layer, **kw)
        torch.cuda.synchronize(); res[name] = time.perf_counter() - t0
    # torch autograd + torch.optim.Adam over the same HIP layer (the straightforward port of the reference loop)
    target = pred.cuda() * 1000.0
    def autograd_fit():
        shape, rot, pose = (torch.zeros(B, n, device="cuda", requires_grad=True) for n in (10, 3, 45))
        trans = (target.mean(1) / 1000.0).clone().requires_grad_()
        mse = torch.nn.MSELoss()
        for group, lr, n in (([rot, trans], 1e-1, 500), ([rot, pose, shape, trans], 1e-2, 700)):
            opt = torch.optim.Adam(group, lr=lr)
            for _ in range(n):
                loss = mse(layer(torch.cat((rot, pose), 1), shape, trans)[0], target)
                opt.zero_grad(); loss.backward(); opt.step()
        return loss.item()
    autograd_fit()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    l = autograd_fit()
    torch.cuda.synchronize(); res["autograd+torch Adam"] = time.perf_counter() - t0
    err = float(((torch.from_numpy(out["verts"]) - pred * 1000.0) ** 2).mean())
    print("B=%5d  " % B + "  ".join("%s %.3f s" % kv for kv in res.items()) + "   final MSE %.4f / %.4f mm^2" % (err, l), flush=True)
