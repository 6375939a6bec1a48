"""The four live helpers of utils/opt_utils.py (:7-45): min-max colour scalers (torch only, no sklearn)."""
import torch


def _minmax(x):
    lo, hi = x.min(0, keepdim=True)[0], x.max(0, keepdim=True)[0]
    d = hi - lo
    d[d == 0.] = 1.
    return (x - lo) / d


def get_mano_vert_colors(mano_layer):
    dev = mano_layer.th_faces.device if mano_layer.th_faces.is_cuda else "cuda"
    verts, _ = mano_layer(torch.zeros(1, 48, device=dev), torch.zeros(1, 10, device=dev), torch.zeros(1, 3, device=dev))
    return _minmax(verts[0].detach().cpu()).numpy()


def get_upscale_mano_vert_colors(upscale_vertices):
    return _minmax(torch.as_tensor(upscale_vertices)).numpy()


class PyTMinMaxScaler(object):
    """Transforms each channel to the range [0, 1] (utils/opt_utils.py:32-41)."""
    def __call__(self, tensor):
        dist = (tensor.max(dim=1, keepdim=True)[0] - tensor.min(dim=1, keepdim=True)[0])
        dist[dist == 0.] = 1.
        scale = 1.0 / dist
        tensor.mul_(scale).sub_(tensor.min(dim=1, keepdim=True)[0])
        return tensor


def scale_value(tensor):
    return PyTMinMaxScaler()(tensor)
