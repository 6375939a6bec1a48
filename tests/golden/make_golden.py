"""Generate golden input/output vectors by IMPORTING THE REFERENCE in the build container.

    python tests/golden/make_golden.py          # needs /root/reference (absent on the GPU box)

Writes small .npz fixtures (data only) next to this script:
  mano_layer.npz   ManoLayer.forward (manopth/manolayer.py:108-296) on the synthetic MANO-shaped model
  losses.npz       kps_loss (loss/kps_loss.py), arap_loss (loss/arap.py), albedo_reg / normal_reg
                   (loss/texture_reg.py, CPU RNG seeded; the drawn integer offsets are stored too)
  smooth.npz       LossSmoothPoses.smooth_pose / LossSmoothRoots.smooth_root (loss/smooth.py) through the reference ManoLayer,
                   values + gradients w.r.t. rot / pose / shape / trans / cam
The reference modules that need PyTorch3D cannot be imported and are not covered.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, "/root/reference")

from harp_amd import synth  # noqa: E402


def ref_mano_layer(model):
    # mano.webuser needs chumpy + the licensed pkl: stub the loader module, build the layer by hand
    import types
    stub = types.ModuleType("mano.webuser.smpl_handpca_wrapper_HAND_only")
    stub.ready_arguments = lambda *a, **k: None
    sys.modules.setdefault("mano", types.ModuleType("mano"))
    sys.modules.setdefault("mano.webuser", types.ModuleType("mano.webuser"))
    sys.modules["mano.webuser.smpl_handpca_wrapper_HAND_only"] = stub
    from manopth.manolayer import ManoLayer
    layer = ManoLayer.__new__(ManoLayer)
    torch.nn.Module.__init__(layer)
    layer.center_idx, layer.robust_rot, layer.rot = None, False, 3
    layer.flat_hand_mean, layer.side, layer.use_pca = False, "right", False
    layer.joint_rot_mode, layer.root_rot_mode, layer.ncomps = "axisang", "axisang", 45
    layer.register_buffer("th_betas", torch.zeros(1, 10))
    layer.register_buffer("th_shapedirs", torch.from_numpy(model["shapedirs"]))
    layer.register_buffer("th_posedirs", torch.from_numpy(model["posedirs"]))
    layer.register_buffer("th_v_template", torch.from_numpy(model["v_template"]).unsqueeze(0))
    layer.register_buffer("th_J_regressor", torch.from_numpy(model["J_regressor"]))
    layer.register_buffer("th_weights", torch.from_numpy(model["weights"]))
    layer.register_buffer("th_faces", torch.from_numpy(model["faces"]).long())
    layer.register_buffer("th_hands_mean", torch.from_numpy(model["hands_mean"]).unsqueeze(0))
    return layer


class DuckMeshes:
    """The subset of pytorch3d.structures.Meshes that loss/arap.py:25-41 touches."""
    def __init__(self, verts, edges):
        self.v, self.e = verts, edges                      # (N,V,3), (E,2)
        self.device = verts.device
    def isempty(self): return False
    def __len__(self): return self.v.shape[0]
    def extend(self, n): return DuckMeshes(self.v.repeat(n, 1, 1), self.e)
    def verts_packed(self): return self.v.reshape(-1, 3)
    def edges_packed(self):
        N, V = self.v.shape[:2]
        return torch.cat([self.e + i * V for i in range(N)], 0)
    def edges_packed_to_mesh_idx(self): return torch.arange(len(self)).repeat_interleave(self.e.shape[0])
    def num_edges_per_mesh(self): return torch.full((len(self),), self.e.shape[0], dtype=torch.int64)


def main():
    tpl = synth.load_template("hand")
    model = synth.make_mano_model(tpl, seed=0)
    topo = synth.build_topology(tpl["faces0"], 778)
    g = torch.Generator().manual_seed(0)
    B = 6
    pose = torch.randn(B, 48, generator=g) * 0.3
    betas = torch.randn(B, 10, generator=g) * 0.5
    trans = torch.randn(B, 3, generator=g) * 0.05
    layer = ref_mano_layer(model)
    pose.requires_grad_(True); betas.requires_grad_(True); trans.requires_grad_(True)
    verts, joints = layer(pose, betas, trans)
    wv = torch.randn(verts.shape, generator=g); wj = torch.randn(joints.shape, generator=g)
    ((verts * wv).sum() + (joints * wj).sum()).backward()
    verts0, joints0 = layer(pose.detach(), betas.detach(), torch.zeros(B, 3))     # trans-ignored branch (:281)
    np.savez_compressed(os.path.join(HERE, "mano_layer.npz"), pose=pose.detach().numpy(), betas=betas.detach().numpy(),
                        trans=trans.detach().numpy(), verts=verts.detach().numpy(), joints=joints.detach().numpy(),
                        wv=wv.numpy(), wj=wj.numpy(), g_pose=pose.grad.numpy(), g_betas=betas.grad.numpy(),
                        g_trans=trans.grad.numpy(), verts_notrans=verts0.numpy(), joints_notrans=joints0.numpy())

    from loss.kps_loss import kps_loss
    from loss.arap import arap_loss
    from loss.texture_reg import albedo_reg, normal_reg
    gt = torch.randn(B, 21, 3, generator=g) * 40.0
    pr = (gt / 1000.0 + torch.randn(B, 21, 3, generator=g) * 0.004).requires_grad_(True)
    lk = kps_loss(gt, pr, use_arm=False, device="cpu"); lk.backward()
    pr22 = torch.cat([pr.detach(), torch.randn(B, 1, 3, generator=g)], 1)
    lk_arm = kps_loss(gt, pr22, use_arm=True, device="cpu")
    V = topo["n_verts"]
    edges = torch.from_numpy(topo["edges"]).long()
    ref_v = torch.randn(1, V, 3, generator=g) * 0.05
    cur_v = (ref_v.repeat(3, 1, 1) + torch.randn(3, V, 3, generator=g) * 0.002).requires_grad_(True)
    la = arap_loss(DuckMeshes(cur_v, edges), DuckMeshes(ref_v, edges)); la.backward()
    tex = (torch.rand(1, 64, 48, 3, generator=g)).requires_grad_(True)
    nm = torch.nn.functional.normalize(torch.randn(1, 64, 48, 3, generator=g) * 0.2 + torch.tensor([0., 0., 1.]), dim=-1).requires_grad_(True)
    mask = (torch.rand(64, 48, generator=g) > 0.4).double()
    torch.manual_seed(123); lt = albedo_reg(tex, uv_mask=mask.clone(), std=1.0); lt.backward()
    torch.manual_seed(123); d1 = torch.normal(mean=0, std=1.0, size=(64, 48, 2)).to(torch.int)
    torch.manual_seed(321); ln = normal_reg(nm, uv_mask=mask.clone()); ln.backward()
    torch.manual_seed(321); d2 = torch.normal(mean=0, std=2.0, size=(64, 48, 2)).to(torch.int)
    np.savez_compressed(os.path.join(HERE, "losses.npz"),
                        kps_gt=gt.numpy(), kps_pred=pr.detach().numpy(), kps_loss=lk.item(), kps_grad=pr.grad.numpy(),
                        kps_pred22=pr22.numpy(), kps_loss_arm=lk_arm.item(),
                        arap_ref=ref_v.numpy(), arap_cur=cur_v.detach().numpy(), arap_loss=la.item(), arap_grad=cur_v.grad.numpy(),
                        tex=tex.detach().numpy(), nm=nm.detach().numpy(), mask=mask.numpy(),
                        albedo_dist=d1.numpy(), albedo_loss=lt.item(), albedo_grad=tex.grad.numpy(),
                        normal_dist=d2.numpy(), normal_loss=ln.item(), normal_grad=nm.grad.numpy())
    # ---- temporal smoothness terms (loss/smooth.py:29-131; dead in the main loop, used by the preprocessing fit — SURVEY §8f rank 4)
    from loss.smooth import LossSmoothPoses, LossSmoothRoots
    T, nF = 12, 6                                         # two "sequences" of 6 frames: clamping at both kinds of boundary
    sp = {"rot": (torch.randn(T, 3, generator=g) * 0.3).requires_grad_(True), "pose": (torch.randn(T, 45, generator=g) * 0.3).requires_grad_(True),
          "shape": (torch.randn(1, 10, generator=g) * 0.5).requires_grad_(True), "trans": (torch.randn(T, 3, generator=g) * 0.02).requires_grad_(True),
          "cam": (torch.tensor([[0.9, 0.02, -0.03]]).repeat(T, 1) + torch.randn(T, 3, generator=g) * 0.02).requires_grad_(True)}
    sfid = torch.tensor([0, 5, 6, 7, 11, 3])
    l_pose = LossSmoothPoses(nF).smooth_pose(sp, sfid, layer, device="cpu")
    l_root = LossSmoothRoots(nF, 1000.0, 224).smooth_root(sp, sfid, layer, device="cpu")
    (l_pose + 1e4 * l_root).backward()
    np.savez_compressed(os.path.join(HERE, "smooth.npz"), fid=sfid.numpy(), n_frames=nF, focal=1000.0, res=224,
                        **{k: v.detach().numpy() for k, v in sp.items()}, smooth_pose=l_pose.item(), smooth_root=l_root.item(),
                        **{"g_" + k: v.grad.numpy() for k, v in sp.items()})
    print("golden fixtures written:", [f for f in os.listdir(HERE) if f.endswith(".npz")])


if __name__ == "__main__":
    main()
