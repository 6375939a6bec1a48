import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from harp_amd import synth, ops
from oracle import harp_ref as H, p3d_like as P
tpl = synth.load_template('hand'); topo_np = synth.build_topology(tpl['faces0'], 778)
model_np = synth.make_mano_model(tpl)
model = {k: torch.from_numpy(v) for k,v in model_np.items()}
topo = {k: torch.from_numpy(np.asarray(v)).long() if isinstance(v,np.ndarray) else v for k,v in topo_np.items()}
T,S=2,256
seq, focal = synth.make_sequence(model_np, T, S)
params = dict(pose=seq['pose'], rot=seq['rot'], trans=seq['trans'], shape=seq['shape'].mean(0), cam=seq['cam'], verts_disps=torch.zeros(3093,1))
fid = torch.arange(T)
j,v = H.prepare_mesh(params, fid, model, topo)
R,Tt = H.camera_RT(params['cam'], S, focal)
_, ndc = P.world_to_ndc(v, R, Tt, focal, (S/2,S/2), S)
ndc = ndc.detach().requires_grad_()
blur, sigma = ops.SIL_BLUR, ops.SIL_SIGMA
p2f, zb, bary, d = P.rasterize_meshes(ndc, topo['faces'], S, blur, 50)
a_ref = P.sigmoid_alpha_blend(p2f, d, sigma)
tgt = (torch.rand(T,S,S) > 0.5).float()
(a_ref - tgt).abs().mean().backward()
g_ref = ndc.grad.clone()
p2f1, zb1, _, _ = P.rasterize_meshes(ndc.detach(), topo['faces'], S, 0.0, 1)
dev = 'cuda'
ndc_d = ndc.detach().to(dev).requires_grad_()
faces_d = topo['faces'].int().to(dev)
alpha, face_id = ops.soft_silhouette(ndc_d, faces_d, S)
(alpha - tgt.to(dev)).abs().mean().backward()
torch.cuda.synchronize()
fid_ref = torch.where(p2f1[...,0] >= 0, p2f1[...,0] % topo['faces'].shape[0], p2f1[...,0]).int()
print('alpha max err', (alpha.cpu()-a_ref).abs().max().item(), 'mismatch>1e-4:', ((alpha.cpu()-a_ref).abs()>1e-4).float().mean().item())
print('face mismatch frac', (face_id.cpu()!=fid_ref).float().mean().item())
g = ndc_d.grad.cpu()
print('grad rel L2', ((g-g_ref).norm()/g_ref.norm()).item(), g_ref.norm().item(), g.norm().item())
f2, z2, _, _ = ops.rasterize_fwd(ndc_d.detach(), faces_d, S, soft=False)
print('hard face mismatch', (f2.cpu()!=fid_ref).float().mean().item(), 'z err', (z2.cpu()-zb1[...,0]).abs().max().item())
# timing
B=32; S2=512
seq2, focal2 = synth.make_sequence(model_np, B, S2)
params2 = dict(pose=seq2['pose'], rot=seq2['rot'], trans=seq2['trans'], shape=seq2['shape'].mean(0), cam=seq2['cam'], verts_disps=torch.zeros(3093,1))
j2,v2 = H.prepare_mesh(params2, torch.arange(B), model, topo)
R2,T2 = H.camera_RT(params2['cam'], S2, focal2)
_, ndc2 = P.world_to_ndc(v2, R2, T2, focal2, (S2/2,S2/2), S2)
nd = ndc2.to(dev).contiguous(); ws = ops.rasterize_workspace(B, faces_d.shape[0], S2, dev)
for soft in (True, False):
    for _ in range(3): ops.rasterize_fwd(nd, faces_d, S2, soft=soft, blur_radius=blur, sigma=sigma, ws=ws)
    torch.cuda.synchronize(); t=time.time()
    for _ in range(20): out = ops.rasterize_fwd(nd, faces_d, S2, soft=soft, blur_radius=blur, sigma=sigma, ws=ws)
    torch.cuda.synchronize(); print('soft' if soft else 'hard', 'B=32 512^2 ms', (time.time()-t)/20*1e3, 'coverage', (out[0]>=0).float().mean().item())
