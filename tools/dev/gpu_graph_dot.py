"""Stream assignment of the step's hipGraph as the runtime instantiates it: run with DEBUG_HIP_GRAPH_DOT_PRINT=1 (the runtime writes
graph_*_dot_print_* into the working directory), then prints node -> stream, which nodes signal, and the edges.
R3_KIND=arm R3_S=1024 selects the C5 workload."""
import sys, os, re, glob; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, bench
S = int(os.environ.get("R3_S", "512")); KIND = os.environ.get("R3_KIND", "hand")
eng, focal = bench.build_engine(0, 1, torch.device('cuda'), T=32, img=S, B=32, kind=KIND)
eng.keep_image = False
eng.set_schedule(torch.arange(32).reshape(1, 32).int())
for _ in range(5): eng.step(None, True, True)
torch.cuda.synchronize()
files = sorted(glob.glob("graph_*_dot_print_*"), key=os.path.getmtime)
if not files:
    sys.exit("no dot file: run with DEBUG_HIP_GRAPH_DOT_PRINT=1")
txt = open(files[-1]).read()
for n, name, sid, sig in re.findall(r'"graph_\d+_node_(\d+)"\[[^\]]*?label="\d+\n(\S+)\nStreamId:(\d+)\nSignalIsRequired: (\w+)', txt):
    short = re.sub(r'^_ZN\d+_GLOBAL__N_1\d+', '', name)
    short = re.sub(r'^_ZN2at6native\d+', 'at::', short)[:40]
    print(f"node {n:>3}  stream {sid}  {'signals' if sig == 'true' else '       '}  {short}")
print("edges:", " ".join(f"{a}>{b}" for a, b in re.findall(r'"graph_\d+_node_(\d+)" -> "graph_\d+_node_(\d+)"', txt)))
