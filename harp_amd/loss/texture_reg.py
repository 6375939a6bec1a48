"""albedo_reg / normal_reg (loss/texture_reg.py:5-66) on the HIP kernels (csrc/losses.hip).  Like the reference, every call
draws fresh integer neighbour offsets with torch.normal on the CPU RNG (texture_reg.py:15, 51), so seeding torch the same
way gives the same loss; pass `dist=` to supply the offsets explicitly (the fused engine draws them on the device)."""
import torch

from .. import _lib


class _SmoothReg(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tex, dist, mask, close_z_scale):
        t = tex.contiguous().float()
        H, W = t.shape[-3], t.shape[-2]
        loss = torch.zeros(1, dtype=torch.float32, device=t.device)
        L = _lib.lib()
        if close_z_scale:
            _lib.check(L.harp_close_to_z_reg(_lib.ptr(t), H, W, close_z_scale, None, _lib.ptr(loss), None, _lib.stream()), "harp_close_to_z_reg")
        _lib.check(L.harp_texture_smooth_reg(_lib.ptr(t), _lib.ptr(dist), _lib.ptr(mask), H, W, None, _lib.ptr(loss), None, _lib.stream()),
                   "harp_texture_smooth_reg")
        ctx.save_for_backward(t, dist, mask)
        ctx.cz = close_z_scale
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        t, dist, mask = ctx.saved_tensors
        H, W = t.shape[-3], t.shape[-2]
        w = g.reshape(1).float().contiguous()
        gt = torch.zeros_like(t)
        scratch = torch.zeros(1, dtype=torch.float32, device=t.device)
        L = _lib.lib()
        if ctx.cz:
            _lib.check(L.harp_close_to_z_reg(_lib.ptr(t), H, W, ctx.cz, _lib.ptr(w), _lib.ptr(scratch), _lib.ptr(gt), _lib.stream()), "harp_close_to_z_reg")
        _lib.check(L.harp_texture_smooth_reg(_lib.ptr(t), _lib.ptr(dist), _lib.ptr(mask), H, W, _lib.ptr(w), _lib.ptr(scratch), _lib.ptr(gt),
                                             _lib.stream()), "harp_texture_smooth_reg")
        return gt, None, None, None


def _offsets(shape_hw, std, device, dist):
    if dist is None:
        dist = torch.normal(mean=0, std=std, size=(*shape_hw, 2)).to(torch.int)                      # texture_reg.py:15 / :51 (CPU RNG)
    return dist.to(torch.int32).contiguous().to(device)


def _mask(uv_mask, device):
    return None if uv_mask is None else uv_mask.to(device=device, dtype=torch.float32).contiguous()


def albedo_reg(uv_texture, std=2.0, uv_mask=None, dist=None):
    """texture_reg.py:5-30; uv_texture (1,H,W,3) on the HIP device."""
    return _SmoothReg.apply(uv_texture, _offsets(uv_texture.shape[-3:-1], std, uv_texture.device, dist), _mask(uv_mask, uv_texture.device), 0.0)


def normal_reg(normal_map, std=2.0, uv_mask=None, dist=None):
    """texture_reg.py:33-37: 0.2 * close_to_z_reg + smooth_texture_reg (close_to_z keeps the reference's norm-over-width quirk)."""
    return _SmoothReg.apply(normal_map, _offsets(normal_map.shape[-3:-1], std, normal_map.device, dist), _mask(uv_mask, normal_map.device), 0.2)
