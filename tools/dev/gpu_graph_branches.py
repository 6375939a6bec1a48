"""How many branches of a captured hipGraph run concurrently?  N streams each run one ~100-us single-workgroup spin kernel; the replay
time is ~100 us if all N overlap, N x 100 if they serialise."""
import time, torch
dev = torch.device("cuda")
cyc = 250000
def run(n):
    streams = [torch.cuda.Stream() for _ in range(n)]
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                torch.cuda._sleep(cyc)
        for s in streams:
            cur.wait_stream(s)
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize(); tg = (time.perf_counter() - t) / 20 * 1e6
    # eager
    cur = torch.cuda.current_stream()
    def eager():
        for s in streams:
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                torch.cuda._sleep(cyc)
        for s in streams:
            cur.wait_stream(s)
    for _ in range(3): eager()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): eager()
    torch.cuda.synchronize(); te = (time.perf_counter() - t) / 20 * 1e6
    return tg, te
for n in (1, 2, 3, 4, 6, 8):
    tg, te = run(n)
    print(f"{n} branches: graph replay {tg:7.1f} us   eager {te:7.1f} us", flush=True)
