"""Would two half-batch graphs replayed concurrently on two streams beat one full-batch graph?  (upper bound for micro-batching)"""
import sys, time; sys.path.insert(0, '.')
import torch, bench
dev = torch.device('cuda')
def mk(B, seed):
    eng, _ = bench.build_engine(0, 1, dev, T=32, img=512, B=B, seed=seed) if 'seed' in bench.build_engine.__code__.co_varnames else bench.build_engine(0, 1, dev, T=32, img=512, B=B)
    return eng
e32 = mk(32, 0)
ea, eb = mk(16, 0), mk(16, 0)
f32 = torch.arange(32, dtype=torch.int32, device=dev)
fa, fb = f32[:16].clone(), f32[16:].clone()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for _ in range(3):
    e32.step(f32, True, True, use_graph=True)
    with torch.cuda.stream(s1): ea.step(fa, True, True, use_graph=True)
    with torch.cuda.stream(s2): eb.step(fb, True, True, use_graph=True)
torch.cuda.synchronize()
def t(fn, n=50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def two():
    with torch.cuda.stream(s1): ea.step(fa, True, True, use_graph=True)
    with torch.cuda.stream(s2): eb.step(fb, True, True, use_graph=True)
def seq():
    ea.step(fa, True, True, use_graph=True); eb.step(fb, True, True, use_graph=True)
print('one graph  B=32        ms/step', t(lambda: e32.step(f32, True, True, use_graph=True)))
print('two graphs B=16 serial ms/pair', t(seq))
print('two graphs B=16 concurrent ms/pair', t(two))
