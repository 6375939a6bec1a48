"""Evaluation metrics of the reference's utils/eval_util.py that do not need absent third-party packages:
silhouette IoU, masked-free L1, white-background fill, Procrustes alignment.  LPIPS / MS-SSIM need `lpips` / `pytorch_msssim`
(not installed, no network): `image_eval` reports them as None.  All metrics run on whatever device the tensors live on."""
import numpy as np
import torch


def fill_bg(img, mask):
    """utils/eval_util.py:29-32: composite (N,H,W,3) over a white background with a (N,H,W) mask"""
    m = mask.unsqueeze(-1)
    return img * m + (m - 1) * -1.0


def l1_diff(ref_image, ref_mask, pred_image, pred_mask):
    """utils/eval_util.py:35-39: mean |ref - pred| over everything, images in [0,1] (the masks are accepted and ignored, as upstream)"""
    return torch.mean(torch.abs(ref_image - pred_image)).detach().cpu().numpy()


def sil_iou(ref_masks, pred_masks):
    """utils/eval_util.py:42-50: per-image IoU of the >= 0.5 masks, averaged over the batch"""
    r, p = ref_masks >= 0.5, pred_masks >= 0.5
    union = torch.logical_or(r, p).sum([1, 2])
    inter = torch.logical_and(r, p).sum([1, 2])
    return torch.mean(inter / union).detach().cpu().numpy()


def image_eval(images_for_eval):
    """utils/eval_util.py:10-26: dict of lists of (n,H,W[,3]) tensors -> {"Silhouette IoU", "L1", "LPIPS", "MS_SSIM"}"""
    ev = {k: torch.vstack(v) for k, v in images_for_eval.items()}
    return {"Silhouette IoU": sil_iou(ev["ref_mask"], ev["pred_mask"]),
            "L1": l1_diff(ev["ref_image"], ev["ref_mask"], ev["pred_image"], ev["pred_mask"]),
            "LPIPS": None, "MS_SSIM": None}


def align_w_scale(mtx1, mtx2, return_trafo=False):
    """utils/eval_util.py:212-235 (FreiHAND-style Procrustes): align mtx2 (K,3) to mtx1 (K,3) by translation, scale and rotation"""
    from scipy.linalg import orthogonal_procrustes
    mtx1, mtx2 = np.asarray(mtx1, dtype=np.float64), np.asarray(mtx2, dtype=np.float64)
    t1, t2 = mtx1.mean(0), mtx2.mean(0)
    a, b = mtx1 - t1, mtx2 - t2
    s1, s2 = np.linalg.norm(a) + 1e-8, np.linalg.norm(b) + 1e-8
    a, b = a / s1, b / s2
    R, s = orthogonal_procrustes(a, b)
    if return_trafo:
        return R, s, s1, t1 - t2
    return np.dot(b, R.T) * s * s1 + t1
