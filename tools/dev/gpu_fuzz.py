"""Randomised parity sweep (more seeds / sizes than the committed tests): rasteriser + soft silhouette + gradients vs the oracle,
and the engine's full-step losses vs the oracle.  Prints the worst figures."""
import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
from tests._scene import make_scene, oracle_params, rel
from harp_amd import ops
from harp_amd.engine import FitEngine, LOSS_NAMES
from oracle import harp_ref as H, p3d_like as P
DEV = 'cuda'
worst = dict(alpha_frac=0, face_frac=0, grad_rel=0, z=0, loss_rel=0, ggrad=0)
for seed, S in [(11, 64), (12, 96), (13, 160), (14, 200), (15, 128), (16, 112)]:
    sc = make_scene(T=2, S=S, seed=seed); topo = sc['topo']; focal = sc['focal']
    params = dict(pose=sc['seq']['pose'], rot=sc['seq']['rot'], trans=sc['seq']['trans'], shape=sc['seq']['shape'].mean(0), verts_disps=torch.randn(3093, 1) * 0.001)
    fid = torch.arange(2)
    with torch.no_grad():
        _, v = H.prepare_mesh(params, fid, sc['model'], topo)
        R, T = H.camera_RT(sc['seq']['cam'][fid], S, focal)
        _, ndc = P.world_to_ndc(v, R, T, focal, (S / 2, S / 2), S)
    ndc = ndc.requires_grad_()
    p2f, zb, bary, d = P.rasterize_meshes(ndc, topo['faces'], S, ops.SIL_BLUR, 50)
    a_ref = P.sigmoid_alpha_blend(p2f, d, ops.SIL_SIGMA)
    tgt = (torch.rand(2, S, S) > 0.5).float()
    (a_ref - tgt).abs().mean().backward()
    p2f1, zb1, _, _ = P.rasterize_meshes(ndc.detach(), topo['faces'], S, 0.0, 1)
    fid_ref = torch.where(p2f1[..., 0] >= 0, p2f1[..., 0] % topo['faces'].shape[0], p2f1[..., 0]).int()
    ndc_d = ndc.detach().to(DEV).requires_grad_(); faces_d = topo['faces'].int().to(DEV)
    alpha, face_id = ops.soft_silhouette(ndc_d, faces_d, S)
    (alpha - tgt.to(DEV)).abs().mean().backward()
    f2, z2, _, _ = ops.rasterize_fwd(ndc_d.detach(), faces_d, S, soft=False)
    m = (f2.cpu() == fid_ref)
    r = dict(alpha_frac=((alpha.cpu() - a_ref).abs() > 1e-4).float().mean().item(), face_frac=(face_id.cpu() != fid_ref).float().mean().item(),
             grad_rel=rel(ndc_d.grad.cpu(), ndc.grad).item() if torch.is_tensor(rel(ndc_d.grad.cpu(), ndc.grad)) else rel(ndc_d.grad.cpu(), ndc.grad),
             z=(z2.cpu() - zb1[..., 0])[m].abs().max().item())
    # full step
    eng = FitEngine(sc['model_np'], sc['topo_np'], sc['tpl']['verts_uvs'], sc['tpl']['faces_uvs'], sc['uv_mask'].float(), sc['seq'], S, focal, 2, device=DEV)
    tg = sc['targets']; eng.set_targets(tg['y_true'], tg['y_sil'], tg['y_sil_col'])
    with torch.no_grad():
        eng.params['verts_disps'].copy_(torch.randn(3093, 1) * 0.001); eng.params['texture'].copy_(torch.rand(1, 512, 512, 3) * 0.5 + 0.3)
        eng.params['normal_map'].copy_(torch.tensor([0., 0., 1.]).repeat(1, 512, 512, 1) + torch.randn(1, 512, 512, 3) * 0.1)
    eng.compute_reference_mesh()
    Pp = oracle_params(sc, eng.params)
    f = torch.tensor([1, 0]); eng.fid.copy_(f.int().to(DEV)); eng.tfid.copy_(f.int().to(DEV))
    eng.auto_draw = False; eng.draw_texture_offsets(); eng.set_stage(True, True)
    with torch.no_grad():
        _, rv = H.prepare_mesh(Pp, torch.tensor([0]), sc['model'], sc['topo'])
    loss, total, aux = H.step_losses(Pp, f, sc['model'], sc['topo'], tg, S, focal, rv, eng.dist_albedo.cpu().long(), eng.dist_normal.cpu().long())
    total.backward()
    eng.forward_backward(True, True); torch.cuda.synchronize()
    got = eng.losses()
    r['loss_rel'] = max(abs(got[k] - loss[k].item()) / (abs(loss[k].item()) + 1e-12) for k in LOSS_NAMES if k in loss)
    r['ggrad'] = max(float(rel(eng.grads[k].cpu(), Pp[k].grad)) for k in ('pose', 'cam', 'verts_disps', 'shape', 'light_positions', 'texture', 'normal_map'))
    print(seed, S, {k: f'{v:.2e}' for k, v in r.items()}, flush=True)
    for k in worst: worst[k] = max(worst[k], r[k])
print('WORST', {k: f'{v:.2e}' for k, v in worst.items()})
