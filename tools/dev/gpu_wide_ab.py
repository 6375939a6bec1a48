"""same-box A/B of engine switches on graph-replayed steps (no profiler): python tools/dev/gpu_wide_ab.py [kind img B] ; HARP_AB="wide_front=1,wide_back=1|wide_front=0,wide_back=0"
interleaved fresh captures, 200 replays each, min / median over the rounds"""
import sys, os, time, statistics; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, bench
kind = sys.argv[1] if len(sys.argv) > 1 else "hand"
img = int(sys.argv[2]) if len(sys.argv) > 2 else 512
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
T = 256 if kind == "hand" else 32
T = (T // B) * B if T >= B else B
eng, focal = bench.build_engine(0, 1, torch.device('cuda'), T=T, img=img, B=B, kind=kind)
eng.keep_image = False
eng.set_schedule(torch.arange(T).reshape(-1, B).int())
variants = os.environ.get("HARP_AB", "wide_front=1,wide_back=1|wide_front=0,wide_back=0").split("|")
def run(var, n=200):
    for kv in filter(None, var.split(",")):
        k, v = kv.split("="); setattr(eng, k, type(getattr(eng, k))(int(v)))
    eng._graphs = {}
    for _ in range(8): eng.step(None, True, True)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): eng.step(None, True, True)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
res = {v: [] for v in variants}
R = int(os.environ.get("HARP_AB_ROUNDS", "9"))
for v in variants: run(v, 50)                           # clocks / caches settle
for rep in range(R):
    order = variants[rep % len(variants):] + variants[:rep % len(variants)]      # rotate: no variant always runs first
    for v in order:
        res[v].append(run(v))
base = variants[0]
for v in variants:
    d = [a - b for a, b in zip(res[v], res[base])]
    print(f"{kind} {img} B={B}  {v:48s} min {min(res[v]):.4f}  median {statistics.median(res[v]):.4f} ms  paired vs first: median {statistics.median(d) * 1e3:+.1f} us  [{min(d) * 1e3:+.1f}, {max(d) * 1e3:+.1f}]", flush=True)
