import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from harp_amd import synth
from harp_amd.engine import FitEngine, LOSS_NAMES
from oracle import harp_ref as H, p3d_like as P
torch.manual_seed(0)
tpl = synth.load_template('hand'); topo_np = synth.build_topology(tpl['faces0'], 778)
model_np = synth.make_mano_model(tpl)
model = {k: torch.from_numpy(v) for k,v in model_np.items()}
topo = {k: torch.from_numpy(np.asarray(v)).long() if isinstance(v,np.ndarray) else v for k,v in topo_np.items()}
T,S,B=3,128,2
seq, focal = synth.make_sequence(model_np, T, S)
with torch.no_grad():
    v, j = H.mano_forward(model, torch.cat((seq['rot'], seq['pose']),1), seq['shape'].mean(0).repeat(T,1), seq['trans'])
seq['joints'] = j + torch.randn_like(j)*3.0
uv_mask = torch.from_numpy(tpl['uv_mask']).double()/255
dev='cuda'
eng = FitEngine(model_np, topo_np, tpl['verts_uvs'], tpl['faces_uvs'], uv_mask.float(), seq, S, focal, B, device=dev)
y_true = torch.rand(T,S,S,3); y_sil = (torch.rand(T,S,S)>0.5).float(); y_col = (torch.rand(T,S,S)>0.4).float()
eng.set_targets(y_true, y_sil, y_col)
# perturb params so every gradient path is exercised
with torch.no_grad():
    eng.params['verts_disps'].copy_(torch.randn(3093,1)*0.001)
    eng.params['texture'].copy_(torch.rand(1,512,512,3)*0.5+0.3)
    eng.params['normal_map'].copy_(torch.tensor([0.,0.,1.]).repeat(1,512,512,1)+torch.randn(1,512,512,3)*0.1)
    eng.params['trans'].copy_(torch.randn(T,3)*0.01)
eng.compute_reference_mesh()
# oracle params
Pm = {k: eng.params[k].detach().cpu().clone().requires_grad_() for k in ('pose','cam','verts_disps','shape','light_positions','amb_ratio','texture','normal_map','rot','trans')}
Pm.update(verts_uvs=torch.from_numpy(tpl['verts_uvs']), faces_uvs=torch.from_numpy(tpl['faces_uvs']).long(), uv_mask=uv_mask, init_joints=seq['joints'])
fid = torch.tensor([2,0])
eng.fid.copy_(fid.int().to(dev)); eng.tfid.copy_(fid.int().to(dev))
eng.auto_draw = False; eng.draw_texture_offsets(); eng.set_stage(True, True)
da, dn = eng.dist_albedo.cpu().long(), eng.dist_normal.cpu().long()
with torch.no_grad():
    _, rv = H.prepare_mesh(Pm, torch.tensor([0]), model, topo)
# NOTE: reference mesh was computed with the same perturbed params on both sides
loss, total, aux = H.step_losses(Pm, fid, model, topo, dict(y_true=y_true, y_sil=y_sil, y_sil_col=y_col), S, focal, rv.detach(), da, dn)
total.backward()
eng.forward_backward(True, True)
torch.cuda.synchronize()
lv = eng.losses()
for k in LOSS_NAMES: print(f'{k:14s} oracle {loss[k].item():.6e}  hip {lv[k]:.6e}  rel {abs(lv[k]-loss[k].item())/(abs(loss[k].item())+1e-30):.2e}')
def rel(a,b): return ((a-b).norm()/(b.norm()+1e-30)).item()
print('ref mesh err', (eng.ref_verts.cpu()-rv[0]).abs().max().item())
print('alpha err', (eng.s['alpha'].cpu()-aux['y_sil_pred']).abs().max().item(), 'rgb err frac>1e-4', ((eng.s['rgb'].cpu()-aux['y_pred']).abs().max(-1).values>1e-4).float().mean().item())
for k in ('pose','cam','verts_disps','shape','light_positions','amb_ratio','texture','normal_map','rot','trans'):
    g = eng.grads[k].cpu(); r = Pm[k].grad
    print(f'grad {k:16s} rel {rel(g, r):.3e}  |ref| {r.norm().item():.3e}')
# ---- Adam parity over 3 steps (eager), same offsets
opt_c = torch.optim.Adam([{'params':[Pm['pose'],Pm['cam']],'lr':1e-3},{'params':[Pm['verts_disps'],Pm['shape']],'lr':1e-3}])
opt_a = torch.optim.Adam([Pm['light_positions'],Pm['amb_ratio'],Pm['texture'],Pm['normal_map']], lr=1e-2)
eng.auto_draw = True
for it in range(3):
    fid = torch.tensor([(it)%T, (it+1)%T])
    eng.step(fid, True, True, use_graph=(it>0))
    da, dn = eng.dist_albedo.cpu().long(), eng.dist_normal.cpu().long()
    loss, total, aux = H.step_losses(Pm, fid, model, topo, dict(y_true=y_true, y_sil=y_sil, y_sil_col=y_col), S, focal, rv.detach(), da, dn)
    opt_c.zero_grad(); opt_a.zero_grad()
    for k in ('rot','trans'): Pm[k].grad = None
    total.backward(); opt_c.step(); opt_a.step()
torch.cuda.synchronize()
for k in ('pose','cam','verts_disps','shape','light_positions','amb_ratio','texture','normal_map'):
    a, r = eng.params[k].cpu(), Pm[k].detach()
    d=(a-r).abs(); print(f'param {k:16s} max abs diff {d.max().item():.3e} mean {d.mean().item():.3e} frac>1e-3 {(d>1e-3).float().mean().item():.3e} (step scale ~ lr)')
