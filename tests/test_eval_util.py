"""CPU tests of the metric helpers (harp_amd/utils/eval_util.py)."""
import numpy as np
import torch

from harp_amd.utils import eval_util as E


def test_iou_l1_fill():
    ref = torch.zeros(2, 8, 8); pred = torch.zeros(2, 8, 8)
    ref[0, 2:6, 2:6] = 1.0; pred[0, 4:8, 2:6] = 0.7          # 16 vs 16 px, overlap 8 -> IoU 8/24
    ref[1, :4] = 1.0; pred[1, :4] = 0.6                      # identical -> 1
    assert abs(float(E.sil_iou(ref, pred)) - (8 / 24 + 1.0) / 2) < 1e-6
    a, b = torch.rand(2, 8, 8, 3), torch.rand(2, 8, 8, 3)
    assert abs(float(E.l1_diff(a, ref, b, pred)) - (a - b).abs().mean().item()) < 1e-7
    f = E.fill_bg(a, ref)
    assert torch.equal(f[0, 0, 0], torch.ones(3)) and torch.allclose(f[0, 3, 3], a[0, 3, 3])
    st = E.image_eval({"ref_mask": [ref[:1], ref[1:]], "pred_mask": [pred[:1], pred[1:]], "ref_image": [a[:1], a[1:]], "pred_image": [b[:1], b[1:]]})
    assert abs(float(st["Silhouette IoU"]) - (8 / 24 + 1.0) / 2) < 1e-6 and st["LPIPS"] is None


def test_procrustes_recovers_similarity():
    rng = np.random.default_rng(0)
    x = rng.normal(size=(21, 3))
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] *= -1
    y = 1.7 * x @ q.T + np.array([0.3, -0.2, 0.9])
    assert np.abs(E.align_w_scale(x, y) - x).max() < 1e-6
