import sys; sys.path.insert(0,'.')
import torch
from tests._scene import make_scene
from harp_amd.engine import FitEngine
sc = make_scene(T=3, S=128, seed=0)
eng = FitEngine(sc["model_np"], sc["topo_np"], sc["tpl"]["verts_uvs"], sc["tpl"]["faces_uvs"], sc["uv_mask"].float(), sc["seq"], 128, sc["focal"], 2, device='cuda')
tg = sc["targets"]; eng.set_targets(tg["y_true"], tg["y_sil"], tg["y_sil_col"])
def snap(): return [t.clone() for t in (eng.p_buf, eng.m_buf, eng.v_buf, eng.hyper)]
def restore(s):
    for d, t in zip((eng.p_buf, eng.m_buf, eng.v_buf, eng.hyper), s): d.copy_(t)
fid = torch.tensor([1,2])
st = snap(); gen_state = eng.gen.get_state()
res = {}
for mode in ('eager','graph','graph2','eager2'):
    restore(st); eng.gen.set_state(gen_state)
    eng.step(fid, True, True, use_graph=mode.startswith('graph')); torch.cuda.synchronize()
    res[mode] = (eng.p_buf.clone(), eng.g_buf.clone())
for m in ('graph','graph2','eager2'):
    dp = (res[m][0]-res['eager'][0]).abs(); dg = (res[m][1]-res['eager'][1]).abs()
    print(m, 'param max diff', dp.max().item(), 'grad max diff', dg.max().item())
    for k in ('pose','cam','shape','verts_disps','texture'):
        o,n,_ = eng.arena.offsets[k]
        print('   ', k, 'dp', dp[o:o+n].max().item(), 'dg', dg[o:o+n].max().item(), 'gnorm', res['eager'][1][o:o+n].norm().item())
