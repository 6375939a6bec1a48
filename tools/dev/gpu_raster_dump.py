"""checksums of one forward pass of the bench workload (face ids, depth map, alpha, loss vector) for the library in HARP_LIB_PATH: two
builds whose rasterisers must agree bit for bit print the same line"""
import sys, os, hashlib; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, bench
S = int(os.environ.get("R3_S", "512")); KIND = os.environ.get("R3_KIND", "hand")
eng, focal = bench.build_engine(0, 1, torch.device('cuda'), T=32, img=S, B=32, kind=KIND)
eng.keep_image = True
eng.auto_draw = False
eng.draw_texture_offsets()
eng.step(torch.arange(32), True, True, use_graph=False)
torch.cuda.synchronize()
h = lambda t: hashlib.sha1(t.detach().cpu().numpy().tobytes()).hexdigest()[:12]
s = eng.s
print(os.path.basename(os.environ.get("HARP_LIB_PATH", "default")), "face_c", h(s["face_c"]), "face_l", h(s["face_l"]), "zl", h(s["zl"]), "alpha", h(s["alpha"]),
      "covered", int((s["face_c"] >= 0).sum()), "alpha_sum %.6f" % float(s["alpha"].double().sum()))
if os.environ.get("R3_SAVE"):
    torch.save({"face_c": s["face_c"].cpu(), "ndc": s["ndc_c"][:32].cpu(), "faces": eng.topo.faces.cpu()}, os.environ["R3_SAVE"])
