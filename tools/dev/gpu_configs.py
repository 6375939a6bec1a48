"""frames/s of the engine on the other BASELINE.json configs (parity-test cases, not bench lines): hand C2 at B=18/32, several batch
sizes at 512^2, hand at 1024^2, the SMPL-X arm mesh at 512^2 / 1024^2 (C5's per-GPU share)."""
import sys, os, time; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
from harp_amd import synth
from harp_amd.engine import FitEngine
dev = torch.device('cuda')

def run(mesh, S, B, T, steps=20):
    torch.manual_seed(0)
    if mesh == 'hand':
        tpl = synth.load_template('hand'); topo = synth.build_topology(tpl['faces0'], 778); model = synth.make_mano_model(tpl, seed=0)
        seq, focal = synth.make_sequence(model, T, S, seed=0); seq['joints'] = torch.zeros(T, 21, 3); kw = {}
    else:
        tpl = synth.load_template('arm'); topo = synth.build_topology(tpl['faces0'], 1026); model = synth.make_smplx_arm_model(tpl, seed=0)
        focal = 1000.0 * S / 224.0; g = torch.Generator().manual_seed(1); c = model['v_template'].mean(0)
        seq = dict(pose=torch.randn(T, 45, generator=g) * 0.15, rot=torch.randn(T, 3, generator=g) * 0.2, trans=torch.zeros(T, 3),
                   shape=torch.randn(T, 10, generator=g) * 0.3, joints=torch.zeros(T, 21, 3),
                   cam=torch.tensor([[2 * focal / (S * 1.6), -float(c[0]), -float(c[1])]]).repeat(T, 1))
        kw = dict(use_arm=True, opt_arm_pose=True)
    eng = FitEngine(model, topo, tpl['verts_uvs'], tpl['faces_uvs'], tpl['uv_mask'].astype(np.float32) / 255.0, seq, S, focal, B, device=dev, **kw)
    eng.set_targets(torch.rand(T, S, S, 3), (torch.rand(T, S, S) > 0.5).float(), (torch.rand(T, S, S) > 0.4).float())
    eng.init_joints = torch.zeros(T, eng.n_joints, 3, device=dev)
    sched = torch.stack([(torch.arange(B) + i * B) % T for i in range(8)]).int()
    eng.set_schedule(sched)
    for _ in range(4): eng.step(None, True, True)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(steps): eng.step(None, True, True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / steps
    cov = (eng.s['face_c'] >= 0).float().mean().item()
    ok = all(np.isfinite(v) for v in eng.losses().values())
    print(f'{mesh:5s} S={S:5d} B={B:3d}: {dt*1e3:7.3f} ms/step {B/dt:9.0f} frames/s  coverage {cov:.3f} finite {ok}', flush=True)
    del eng; torch.cuda.empty_cache()

for cfg in [('hand', 512, 18, 72), ('hand', 512, 32, 64), ('hand', 512, 64, 64), ('hand', 512, 128, 128), ('hand', 256, 32, 64), ('hand', 1024, 32, 32),
            ('arm', 512, 32, 32), ('arm', 1024, 32, 32)]:
    run(*cfg)
