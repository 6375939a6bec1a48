import sys, time; sys.path.insert(0,'.')
import numpy as np, torch, torch.nn.functional as Fn
from harp_amd import synth, ops
from oracle import harp_ref as H, p3d_like as P
torch.manual_seed(0)
tpl = synth.load_template('hand'); topo_np = synth.build_topology(tpl['faces0'], 778)
model_np = synth.make_mano_model(tpl)
model = {k: torch.from_numpy(v) for k,v in model_np.items()}
topo = {k: torch.from_numpy(np.asarray(v)).long() if isinstance(v,np.ndarray) else v for k,v in topo_np.items()}
T,S=2,256
seq, focal = synth.make_sequence(model_np, T, S)
dev='cuda'
dt = ops.DeviceTopology(topo_np, tpl['verts_uvs'], tpl['faces_uvs'], dev)
def rel(a,b): return ((a-b).norm()/(b.norm()+1e-30)).item()

# ---- A: mesh ops
v0 = (torch.randn(T,778,3)*0.01 + torch.from_numpy(model_np['v_template'])[None]*1000).requires_grad_()
disp = (torch.randn(3093,1)*0.002).requires_grad_()
vs = torch.cat([v0/1000, (v0/1000)[:, topo['edges0']].mean(2)], 1)
n1 = P.verts_normals(vs, topo['faces']); vd = vs + n1*disp.repeat(T,1,1); n2 = P.verts_normals(vd, topo['faces'])
w1, w2 = torch.randn_like(vd), torch.randn_like(n2)
((vd*w1).sum() + (n2*w2).sum()).backward()
v0d = v0.detach().to(dev).requires_grad_(); dispd = disp.detach().to(dev).requires_grad_()
vs_d = ops.subdivide(v0d, dt, 1.0/1000.0)
n1_d, vd_d = ops.normals_displace(vs_d, dispd, dt)
n2_d = ops.vertex_normals(vd_d, dt)
((vd_d*w1.to(dev)).sum() + (n2_d*w2.to(dev)).sum()).backward()
print('A vd', (vd_d.cpu()-vd).abs().max().item(), 'n2', (n2_d.cpu()-n2).abs().max().item(), 'g_v0 rel', rel(v0d.grad.cpu(), v0.grad), 'g_disp rel', rel(dispd.grad.cpu(), disp.grad))

# ---- B: full render
params = dict(pose=seq['pose'], rot=seq['rot'], trans=seq['trans'], shape=seq['shape'].mean(0), cam=seq['cam'].clone().requires_grad_(),
  verts_disps=torch.zeros(3093,1), texture=(torch.rand(1,512,512,3)*0.5+0.3).requires_grad_(), normal_map=(torch.tensor([0.,0.,1.]).repeat(1,512,512,1)+torch.randn(1,512,512,3)*0.1).requires_grad_(),
  light_positions=torch.tensor(((-0.5,-0.5,-0.5),)).repeat(T,1).requires_grad_(), amb_ratio=torch.tensor(0.4).requires_grad_(),
  verts_uvs=torch.from_numpy(tpl['verts_uvs']), faces_uvs=torch.from_numpy(tpl['faces_uvs']).long())
fid = torch.arange(T)
with torch.no_grad(): _, verts = H.prepare_mesh(params, fid, model, topo)
verts = verts.detach().requires_grad_()
rgb_ref, aux = H.render_rgb(verts, topo, params, params['cam'][fid], S, focal, return_aux=True)
tgt = torch.rand(T,S,S,3); msk = (torch.rand(T,S,S,1)>0.3).float()
(rgb_ref*msk - tgt*msk).abs().mean().backward()
gref = dict(verts=verts.grad, tex=params['texture'].grad, nm=params['normal_map'].grad, lp=params['light_positions'].grad, amb=params['amb_ratio'].grad, cam=params['cam'].grad)

def render_hip(verts, P_, cam):
    B = verts.shape[0]
    R = torch.tensor([[-1.,0,0],[0,-1.,0],[0,0,1.]], device=dev).repeat(B,1,1)
    Tt = torch.stack([-cam[:,1], -cam[:,2], 2*focal/(S*cam[:,0]+1e-9)], 1)
    lp = P_['light_positions'][0].repeat(B,1)
    center = verts.mean(1)
    d = lp - center
    pos = center + d*(1.5/torch.linalg.norm(d, dim=1, keepdim=True))
    z = Fn.normalize(center-pos, eps=1e-5); up = torch.tensor([[0.,1.,0.]], device=dev).expand_as(z)
    x = Fn.normalize(torch.cross(up, z, dim=1), eps=1e-5); y = Fn.normalize(torch.cross(z, x, dim=1), eps=1e-5)
    lR = torch.stack([x,y,z], 1).transpose(1,2)
    lT = -torch.bmm(lR.transpose(1,2), pos[:,:,None])[:,:,0]
    vn = ops.vertex_normals(verts, dt)
    ndc_l = ops.project(verts, lR, lT, focal, S)
    zl, _, _ = ops.depth_raster(ndc_l, dt.faces, S)
    ndc_c = ops.project(verts, R, Tt, focal, S)
    _, face_id, ws = ops.depth_raster(ndc_c, dt.faces, S)
    amb = torch.sigmoid(P_['amb_ratio'])*torch.ones(3, device=dev)
    colors = torch.cat([amb, 1-amb, torch.zeros(3, device=dev)])
    nmap = Fn.normalize(P_['normal_map'][0], dim=-1)
    return ops.shade(ndc_c, verts, vn, P_['texture'][0], nmap, lp, colors, face_id, ws, dt, S, focal, zl=zl, light_R=lR, light_T=lT), zl, face_id

Pd = {k: (v.detach().to(dev).requires_grad_() if v.is_floating_point() else v.to(dev)) for k,v in params.items()}
verts_d = verts.detach().to(dev).requires_grad_()
rgb, zl, face_id = render_hip(verts_d, Pd, Pd['cam'][fid.to(dev)])
(rgb*msk.to(dev) - tgt.to(dev)*msk.to(dev)).abs().mean().backward()
torch.cuda.synchronize()
err = (rgb.cpu()-rgb_ref).abs()
print('B rgb max err', err.max().item(), 'frac>1e-4', (err.max(-1).values>1e-4).float().mean().item(), 'zl err', (zl.cpu()-aux['zbuf_light'][...,0]).abs().max().item())
fidr = torch.where(aux['pix_to_face'][...,0]>=0, aux['pix_to_face'][...,0]%dt.F, aux['pix_to_face'][...,0]).int()
print('  face mismatch', (face_id.cpu()!=fidr).float().mean().item())
g = dict(verts=verts_d.grad, tex=Pd['texture'].grad, nm=Pd['normal_map'].grad, lp=Pd['light_positions'].grad, amb=Pd['amb_ratio'].grad, cam=Pd['cam'].grad)
for k in gref: print('  grad', k, 'rel', rel(g[k].cpu(), gref[k]), 'norm', gref[k].norm().item())
