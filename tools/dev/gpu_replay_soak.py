"""Race detector for the multi-stream step graph: with both learning rates at 0 and a one-row schedule every replay sees the same state, so the
gradient arena and the loss vector of every replay must equal the first one's up to the order of the float atomics (~1e-6); a missing
dependency between two streams would show up as an occasional large difference.  python tools/dev/gpu_replay_soak.py [replays]"""
import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
for keep in (False, True):
    eng, focal = bench.build_engine(0, 1, torch.device('cuda'), T=32, img=512, B=32)
    eng.keep_image = keep
    eng.auto_draw = False                        # (fresh texture-regulariser offsets every step would change the terms)
    eng.draw_texture_offsets()
    eng.set_lr(0.0, 0.0)
    eng.set_schedule(torch.arange(32).reshape(1, 32).int())
    for _ in range(4): eng.step(None, True, True)
    torch.cuda.synchronize()
    g0, l0 = eng.g_buf.double().clone(), eng.loss_vec.double().clone()
    worst_g = worst_l = 0.0
    for i in range(n):
        eng.step(None, True, True)
        if i % 8 == 0 or i == n - 1:
            torch.cuda.synchronize()
            worst_g = max(worst_g, ((eng.g_buf.double() - g0).norm() / g0.norm()).item())
            worst_l = max(worst_l, ((eng.loss_vec.double() - l0).abs() / l0.abs().clamp_min(1e-12)).max().item())
    print(f"keep_image={keep}: {n} replays, worst gradient rel-L2 vs the first replay {worst_g:.2e}, worst loss-term rel diff {worst_l:.2e}, graphs {len(eng._graphs)}", flush=True)
