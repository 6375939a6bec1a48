"""bit-level A/B of two builds of the library on the bench workload: run once per library (HARP_LIB_PATH), each run dumps the rasteriser /
shader outputs and the gradient arena of one forward_backward; `cmp A B` compares two dumps.   python tools/dev/gpu_lib_ab.py dump OUT | cmp A B"""
import sys, os, time; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
if sys.argv[1] == "dump":
    import bench
    eng, focal = bench.build_engine(0, 1, torch.device('cuda'))
    for keep in (True, False):
        eng.keep_image = keep
        fid = torch.arange(32, dtype=torch.int32, device='cuda')
        eng._lane["fid"][:32].copy_(fid); eng._lane["tfid"][:32].copy_(fid)
        eng.auto_draw = False
        eng.forward_backward(True, True)
        torch.cuda.synchronize()
        s = eng.s
        out = {k: s[k].cpu() for k in ("face_c", "face_l", "zl", "alpha", "g_alpha", "g_ndc_c")}
        out["g"] = eng.g_buf.cpu(); out["loss"] = eng._lane["loss_vec"].cpu()
        torch.save(out, sys.argv[2] + (".keep" if keep else ".sparse"))
    eng.set_schedule(torch.arange(256).reshape(-1, 32).int())
    for _ in range(6): eng.step(None, True, True)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(60): eng.step(None, True, True)
    torch.cuda.synchronize(); print(os.path.basename(os.environ.get("HARP_LIB_PATH", "default")), "%.4f ms/step" % ((time.perf_counter() - t) / 60 * 1e3))
else:
    for suf in (".keep", ".sparse"):
        a, b = torch.load(sys.argv[2] + suf), torch.load(sys.argv[3] + suf)
        for k in a:
            x, y = a[k], b[k]
            if suf == ".sparse" and k in ("face_c", "alpha", "g_alpha", "face_l"):
                continue                      # unwritten in empty super-tiles
            if x.dtype == torch.float32:
                print(suf, "%-8s bit-identical %s  max|d| %.3e  differing %d of %d" % (k, torch.equal(x.view(torch.int32), y.view(torch.int32)), (x - y).abs().max().item(), (x != y).sum().item(), x.numel()))
            else:
                print(suf, "%-8s identical %s  mismatches %d of %d" % (k, torch.equal(x, y), (x != y).sum().item(), x.numel()))
