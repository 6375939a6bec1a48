import sys; sys.path.insert(0,'.')
import torch
from tests._scene import make_scene
from harp_amd.engine import FitEngine
sc = make_scene(T=3, S=128, seed=0)
eng = FitEngine(sc["model_np"], sc["topo_np"], sc["tpl"]["verts_uvs"], sc["tpl"]["faces_uvs"], sc["uv_mask"].float(), sc["seq"], 128, sc["focal"], 2, device='cuda')
tg = sc["targets"]; eng.set_targets(tg["y_true"], tg["y_sil"], tg["y_sil_col"])
with torch.no_grad():
    eng.params["verts_disps"].copy_(torch.randn(3093, 1) * 0.001)
    eng.params["texture"].copy_(torch.rand(1, 512, 512, 3) * 0.5 + 0.3)
    eng.params["normal_map"].copy_(torch.tensor([0., 0., 1.]).repeat(1, 512, 512, 1) + torch.randn(1, 512, 512, 3) * 0.1)
    eng.params["trans"].copy_(torch.randn(3, 3) * 0.01)
eng.compute_reference_mesh()
fid = torch.tensor([2,0]).int().cuda(); eng.fid.copy_(fid); eng.tfid.copy_(fid); eng.auto_draw = False; eng.draw_texture_offsets(); eng.set_stage(True, True)
gs = []
for i in range(4):
    eng.forward_backward(True, True); torch.cuda.synchronize()
    gs.append({k: v.clone() for k,v in eng.grads.items()})
    print(i, {k: float((gs[i][k]-gs[0][k]).abs().max()) for k in ('pose','cam','shape','texture')}, 'nan', any(torch.isnan(v).any().item() for v in gs[i].values()))
    print('   scratch nan:', {k: bool(torch.isnan(v).any()) for k,v in eng.s.items() if v.dtype==torch.float32 and torch.isnan(v).any()})
p0 = eng.p_buf.clone()
for it in range(3):
    eng.step(torch.tensor([it%3,(it+1)%3]), True, True, use_graph=(it>0)); torch.cuda.synchronize()
    print('  nan grads:', [k for k,v in eng.grads.items() if torch.isnan(v).any()], 'nan scratch:', [k for k,v in eng.s.items() if v.dtype==torch.float32 and torch.isnan(v).any()])
    print('step', it, 'param nan', bool(torch.isnan(eng.p_buf).any()), 'grad nan', bool(torch.isnan(eng.g_buf).any()), 'max |dp|', float((eng.p_buf-p0).abs().max()),
          'hyper', eng.hyper.cpu().numpy().view(eng.hyper_np.dtype))
