"""time harp_conv3x3 at the ten VGG16 shapes (and their data gradients): TFLOP/s per layer, both arithmetic modes.
   python tools/dev/conv_bench.py [N] [S]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from harp_amd.model import conv_hip as C

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = "cuda:0"
ONLY = os.environ.get("CONV_LAYER")        # e.g. "256,256,4": one layer (Cin, Cout, image side divisor)
PRECS = [int(p) for p in os.environ.get("CONV_PREC", "0,1").split(",")]
LAYERS = [(16, 64, S), (64, 64, S), (64, 128, S // 2), (128, 128, S // 2), (128, 256, S // 4), (256, 256, S // 4), (256, 512, S // 8), (512, 512, S // 8)]
if ONLY:
    ci, co, d = (int(v) for v in ONLY.split(","))
    LAYERS = [(ci, co, S // d)]
for prec in PRECS:
    tot_t = tot_f = 0.0
    for Cin, Cout, s in LAYERS:
        x = torch.randn(N, s, s, Cin, device=dev)
        w = torch.randn(Cout, Cin, 3, 3, device=dev) * (2.0 / (9 * Cin)) ** 0.5
        b = torch.zeros(Cout, device=dev)
        f = C.pack_filters(w, prec)
        out = torch.empty(N, s, s, Cout, device=dev)
        for _ in range(2):
            C.conv3x3(x, f, Cout, bias=b, precision=prec, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps):
            C.conv3x3(x, f, Cout, bias=b, precision=prec, out=out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        fl = 2.0 * 9 * Cin * Cout * s * s * N
        tot_t += ms; tot_f += fl
        print(f"prec {prec}  {Cin:4d}->{Cout:4d} @ {s:4d}^2 x{N}: {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s", flush=True)
    print(f"prec {prec}  sum {tot_t:.2f} ms, {tot_f / tot_t / 1e9:.1f} TFLOP/s")
