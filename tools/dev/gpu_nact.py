"""how many super-tiles / tiles hold work in the bench workload (camera and light view)"""
import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, bench
eng, _ = bench.build_engine(0, 1, torch.device('cuda'), T=32)
eng.step(torch.arange(32), True, True, use_graph=False); torch.cuda.synchronize()
B, S = 32, eng.S
nst = ((S + 63) // 64) ** 2
for name in ("ws_c", "ws_l"):
    ws = eng.s[name]
    nact = int(ws[-256:].view(torch.int32)[0])
    print(name, "non-empty super-tiles", nact, "of", B * nst, "-> active workgroups", nact * 16, "of", B * nst * 16)
for name in ("face_c", "face_l"):
    f = eng.s[name]
    t = (f.view(B, S // 16, 16, S // 16, 16) >= 0).any(4).any(2)
    print(name, "tiles with a covered pixel", int(t.sum()), " covered pixels %.3f" % (f >= 0).float().mean().item(), " pixels per covered tile %.1f" % ((f >= 0).sum().item() / max(int(t.sum()), 1)))
