"""Upper bounds: the graph-replayed step with one phase of the shader backward switched off (harp_shade_args.debug_skip bits 8-13: results
WRONG, timing only) — what a restructuring of that phase could gain at most."""
import sys, os, time; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, bench
eng, focal = bench.build_engine(0, 1, torch.device('cuda'))
eng.keep_image = False
eng.set_schedule(torch.arange(256).reshape(-1, 32).int())
orig = eng._shade_struct
for name, bits in (("full", 0), ("no shadow window", 512), ("no vertex phase", 1024), ("no texel records", 256), ("no shadow window, no vertex phase", 1536), ("full", 0)):
    def patched(B, app, bits=bits):
        a = orig(B, app); a.debug_skip = bits; return a
    eng._shade_struct = patched
    eng._graphs = {}
    res = []
    for rep in range(3):
        for _ in range(6): eng.step(None, True, True)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(60): eng.step(None, True, True)
        torch.cuda.synchronize(); res.append((time.perf_counter() - t) / 60 * 1e3)
    print("%-36s" % name, " ".join("%.4f" % r for r in res), "ms/step")
