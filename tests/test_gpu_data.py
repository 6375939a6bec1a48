"""-m gpu: the input wire format end to end: the committed 6-frame fixture tree (tests/golden/frames, expected
tensors from the reference's load_img restated with cv2.erode written out from the OpenCV documentation:
tests/golden/make_golden_frames.py) -> load_multiple_sequences -> ResidentTargets on the device -> FitEngine.set_targets -> steps."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_resident_targets_from_the_committed_frames():
    from harp_amd import synth
    from harp_amd.engine import FitEngine
    from harp_amd.utils import data_util as D
    exp = np.load(os.path.join(GOLDEN, "frames_expected.npz"))
    root = os.path.join(GOLDEN, "frames")
    mp, ds, _, _ = D.load_multiple_sequences(os.path.join(root, "metro"), os.path.join(root, "img"), train_list=["1"], val_list=[])
    rt = D.ResidentTargets(ds, device=DEV)
    y_true, y_sil, y_col = rt.tensors()
    assert y_true.is_cuda and y_true.shape == (6, 64, 64, 3) and y_sil.shape == (6, 64, 64) and y_col.shape == (6, 64, 64)
    assert np.array_equal(y_true.cpu().numpy(), exp["rgb"]) and np.array_equal(y_sil.cpu().numpy(), exp["mask"][..., 0])
    assert np.array_equal(y_col.cpu().numpy(), exp["eroded"])
    # a shard in another order (what a rank of an N > 1 job holds): rows follow `frames`
    sh = D.ResidentTargets(ds, frames=[4, 2], device=DEV)
    assert sh.fid.tolist() == [4, 2] and torch.equal(sh.y_sil_col[0], y_col[4]) and torch.equal(sh.y_true[1], y_true[2])
    # the engine keeps exactly these tensors resident and steps on them (METRO parameters of the same fixture)
    tpl = synth.load_template("hand")
    model = synth.make_mano_model(tpl, seed=0)
    seq = {k: mp[k].float() for k in ("pose", "rot", "trans", "shape", "cam", "joints")}
    eng = FitEngine(model, synth.build_topology(tpl["faces0"], 778), tpl["verts_uvs"], tpl["faces_uvs"], tpl["uv_mask"].astype(np.float32) / 255.0,
                    seq, 64, 1000.0 * 64 / 224.0, 3, device=DEV)
    eng.set_targets(*rt.tensors())
    assert eng.y_true.data_ptr() == y_true.data_ptr() or torch.equal(eng.y_true, y_true)
    assert torch.allclose(eng.bg_sil.sum(), y_sil.sum(), rtol=1e-5)          # per-super-tile background table of the silhouette term
    eng.keep_image = False
    for it in range(3):
        eng.step(torch.tensor([it % 6, (it + 1) % 6, (it + 2) % 6]), True, True)
    torch.cuda.synchronize()
    lv = eng.losses()
    assert all(np.isfinite(v) for v in lv.values()) and lv["silhouette"] > 0 and torch.isfinite(eng.p_buf).all()
