"""arm (SMPL-X) engine: loss-only mode vs image mode give the same losses / gradients"""
import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
from harp_amd import synth
from harp_amd.engine import FitEngine
dev = torch.device('cuda'); S, B, T = 256, 4, 4
tpl = synth.load_template('arm'); topo = synth.build_topology(tpl['faces0'], 1026); model = synth.make_smplx_arm_model(tpl, seed=0)
focal = 1000.0 * S / 224.0; g = torch.Generator().manual_seed(1); c = model['v_template'].mean(0)
seq = dict(pose=torch.randn(T, 45, generator=g) * 0.15, rot=torch.randn(T, 3, generator=g) * 0.2, trans=torch.zeros(T, 3),
           shape=torch.randn(T, 10, generator=g) * 0.3, joints=torch.zeros(T, 21, 3),
           cam=torch.tensor([[2 * focal / (S * 1.6), -float(c[0]), -float(c[1])]]).repeat(T, 1))
res = {}
for keep in (True, False):
    eng = FitEngine(model, topo, tpl['verts_uvs'], tpl['faces_uvs'], tpl['uv_mask'].astype(np.float32) / 255.0, seq, S, focal, B, device=dev,
                    use_arm=True, opt_arm_pose=True)
    gg = torch.Generator().manual_seed(3)
    eng.set_targets(torch.rand(T, S, S, 3, generator=gg), (torch.rand(T, S, S, generator=gg) > 0.5).float(), (torch.rand(T, S, S, generator=gg) > 0.4).float())
    eng.init_joints = torch.zeros(T, eng.n_joints, 3, device=dev)
    eng.keep_image = keep
    fid = torch.arange(B); eng.fid.copy_(fid.int().to(dev)); eng.tfid.copy_(fid.int().to(dev))
    eng.auto_draw = False; eng.draw_texture_offsets(); eng.set_stage(True, True)
    eng.forward_backward(True, True); torch.cuda.synchronize()
    res[keep] = (eng.losses(), eng.g_buf.clone())
for k, v in res[True][0].items():
    assert abs(v - res[False][0][k]) <= 2e-6 * abs(v) + 1e-12, (k, v, res[False][0][k])
d = (res[True][1] - res[False][1]).norm() / res[True][1].norm()
print("arm: losses equal, gradient rel diff %.2e" % d.item()); assert d < 1e-5
eng.set_schedule(torch.arange(T).reshape(1, B))
for _ in range(5): eng.step(None, True, True)
torch.cuda.synchronize(); print("arm steps finite:", bool(torch.isfinite(eng.p_buf).all()))
