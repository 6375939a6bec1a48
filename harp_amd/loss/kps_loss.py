"""kps_loss (loss/kps_loss.py:4-17) on the HIP kernel (csrc/losses.hip: kps_kernel)."""
import torch

from .. import _lib


class _Kps(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gt, pred):
        gt, pred = gt.contiguous().float(), pred.contiguous().float()
        B, NJ = pred.shape[0], pred.shape[1]
        loss = torch.zeros(1, dtype=torch.float32, device=pred.device)
        _lib.check(_lib.lib().harp_kps_loss(_lib.ptr(gt), None, _lib.ptr(pred), B, NJ, None, _lib.ptr(loss), None, _lib.stream()), "harp_kps_loss")
        ctx.save_for_backward(gt, pred)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        gt, pred = ctx.saved_tensors
        B, NJ = pred.shape[0], pred.shape[1]
        w = g.reshape(1).float().contiguous()
        gp = torch.zeros_like(pred)
        scratch = torch.zeros(1, dtype=torch.float32, device=pred.device)
        _lib.check(_lib.lib().harp_kps_loss(_lib.ptr(gt), None, _lib.ptr(pred), B, NJ, _lib.ptr(w), _lib.ptr(scratch), _lib.ptr(gp), _lib.stream()),
                   "harp_kps_loss")
        return None, gp


def kps_loss(gt_kps, pred_kps, use_arm=False, device="cuda"):
    """gt (B,21,3) mm, pred (B,21|22,3) m. With use_arm the first 21 predicted joints are used (kps_loss.py:7-8): the kernel
    reads joints 0..20 of each row, so the slice is implicit."""
    return _Kps.apply(gt_kps.to(pred_kps.device), pred_kps)
