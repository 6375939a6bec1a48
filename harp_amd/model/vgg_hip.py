"""The perceptual term on the HIP convolution kernels (csrc/conv.hip): `L1Loss(vgg(y_pred * mask), vgg(y_true * mask))` of
optimize_sequence.py:546-547 with `vgg` = model/vgg.py's Vgg16Features, forward and data gradient, as `harp_vgg16_term` runs it.

`Vgg16Hip(module)` takes the filters of a `harp_amd.model.vgg.Vgg16Features` (same state-dict keys as the reference module) and packs
them for the matrix cores; nothing here runs a convolution through torch."""
import ctypes

import torch

from .. import _lib
from . import conv_hip
from .vgg import _CONVS, _SLICES, feature_length

_ORDER = sorted(_CONVS)                      # 0, 2, 5, 7, 10, 12, 14, 17, 19, 21: vgg16.features indices of the ten convolutions


def tap_shapes(S):
    """(H, W, C) of the four tap activations relu1_2 ... relu4_3 for an S x S image, NHWC"""
    return [(S, S, 64), (S // 2, S // 2, 128), (S // 4, S // 4, 256), (S // 8, S // 8, 512)]


_DIV = (1, 1, 2, 2, 4, 4, 4, 8, 8, 8)
_COUT = (64, 64, 128, 128, 256, 256, 256, 512, 512, 512)
TAPS = (1, 3, 6, 9)                           # slots of relu1_2, relu2_2, relu3_3, relu4_3 in the 13-slot activation list
# slot of the activation convolution k reads (k = 1..9): relu(conv k-1), or the pooled map behind a tap layer (slots 10..12)
INPUT_SLOT = {1: 0, 2: 10, 3: 2, 4: 11, 5: 4, 6: 5, 7: 12, 8: 7, 9: 8}


def activation_shapes(S):
    """(H, W, C) of the 13 activation slots of harp_vgg16_features: relu(conv 0..9), then the three pooled maps"""
    return [(S // d, S // d, c) for d, c in zip(_DIV, _COUT)] + [(S // 2, S // 2, 64), (S // 4, S // 4, 128), (S // 8, S // 8, 256)]


TILE_SIDES = (16, 8, 8, 8)                    # side of a tile of active_tiles() per resolution level (S/2 at 8 too: -1.3 % of the term's time)


def active_tiles(mask, shift_grid=True, tile_sides=TILE_SIDES):
    """The tiles of each resolution level (side S >> L, L = 0..3) in which activations of image * mask can differ between two images
    sharing `mask` (T,S,S): the support of the mask grown by the receptive field of the level's last convolution — two 3x3 convolutions at
    levels 0 and 1, three at levels 2 and 3, a 2x2 max pool between levels (model/vgg.py:26-33).  Tiles are tile_sides[L] pixels square:
    16 (one workgroup of csrc/conv.hip) at full resolution, 8 (one WAVE of a workgroup) below, where a hand is a few tiles across and a
    16-pixel grid spends its tiles on the rim (level 0 stays at 16: the image-side kernels sit on that grid).  shift_grid: every
    frame's tile grid is shifted by the EVEN origin (oy, ox) in [0, side - 2]^2 that needs the fewest tiles (tile (ty, tx) = pixels
    [side ty - oy, +side) x [side tx - ox, +side)): the tiles hug the region instead of straddling it.
    Returns per level: tiles (T, G*G) int32 0/1 with G = ceil(S_L / side) + 1, tile_list (T, max) int32 (ty * G + tx, raster order),
    tile_count (T) int32, max, origin (T,2) int32, G, side."""
    F = torch.nn.functional
    d = (mask > 0).float()[:, None]
    out = []
    for level, grow in enumerate((2, 2, 3, 3)):
        side = int(tile_sides[level])
        if side not in (8, 16) or (level == 0 and side != 16) or (level and side > int(tile_sides[level - 1])):
            raise ValueError("active_tiles: tile sides are 8 or 16 pixels, 16 at level 0, and no 16 at a coarser level than an 8")
        if level:
            d = F.max_pool2d(d, 2, 2)
        d = F.max_pool2d(d, 2 * grow + 1, 1, grow)
        T, s = d.shape[0], d.shape[-1]
        G = (s + side - 1) // side + 1
        h = side // 2                                                       # even origins: work on 2x2 blocks, tiles of side / 2 blocks
        d2 = F.max_pool2d(d, 2, 2, ceil_mode=True)
        best_n = torch.full((T,), 1 << 30, dtype=torch.int64, device=d.device)
        best_t = torch.zeros(T, G, G, dtype=torch.bool, device=d.device)
        best_o = torch.zeros(T, 2, dtype=torch.int32, device=d.device)
        for oy in (range(h) if shift_grid else (0,)):
            for ox in (range(h) if shift_grid else (0,)):
                p = F.pad(d2, (ox, h * G - ox - d2.shape[-1], oy, h * G - oy - d2.shape[-2]))
                t = F.max_pool2d(p, h, h)[:, 0] > 0                         # (T, G, G)
                n = t.reshape(T, -1).sum(1)
                better = n < best_n
                best_n = torch.where(better, n, best_n)
                best_t = torch.where(better[:, None, None], t, best_t)
                best_o = torch.where(better[:, None], torch.tensor([2 * oy, 2 * ox], dtype=torch.int32, device=d.device)[None], best_o)
        flat = best_t.reshape(T, G * G)
        count = flat.sum(1).int()
        mx = max(int(count.max()), 1)
        order = torch.argsort((~flat).int(), dim=1, stable=True)[:, :mx].int()      # active tiles first, in raster order
        out.append((flat.int().contiguous(), order.contiguous(), count.contiguous(), mx, best_o.contiguous(), G, side))
    return out


def active_share(bound, S):
    """share of each level's image area inside the active tiles of active_tiles()"""
    return [float(b[2].float().sum()) * b[6] ** 2 / (b[0].shape[0] * float(S >> lv) ** 2) for lv, b in enumerate(bound)]


class Vgg16Hip:
    def __init__(self, module, device, precision=conv_hip.F32):
        self.dev = torch.device(device)
        self.precision = int(precision)
        self.layers_weights = [float(w) for w in module.layers_weights]
        sd = module.state_dict()
        slice_of = {ix: n for n, (lo, hi) in enumerate(_SLICES, start=1) for ix in range(lo, hi)}
        self._keep = []                          # device tensors the C struct points into
        net = _lib.Vgg16()
        self.packed = []                         # per convolution: (filters, data-gradient filters, bias) — what the module-level call runs on
        for k, ix in enumerate(_ORDER):
            w = sd[f"slice{slice_of[ix]}.{ix}.weight"].detach().to(self.dev).float().contiguous()
            b = sd[f"slice{slice_of[ix]}.{ix}.bias"].detach().to(self.dev).float().contiguous()
            f = conv_hip.pack_filters(w, self.precision)
            ft = conv_hip.pack_filters(w, self.precision, transpose=True) if k > 0 else None
            self._keep += [w, b, f, ft]
            self.packed.append((f, ft, b))
            if k == 0:
                self._w0 = w
            net.filters[k], net.filters_t[k], net.bias[k] = _lib.ptr(f), _lib.ptr(ft), _lib.ptr(b)
            if k == 0:      # (64,3,3,3) -> (9,3,64) with the taps mirrored: the layout the last backward kernel reads with scalar loads
                w0t = w.reshape(64, 3, 9).flip(2).permute(2, 1, 0).contiguous()
                self._keep.append(w0t)
                net.w0t = _lib.ptr(w0t)
        for i, w in enumerate(self.layers_weights):
            net.layer_w[i] = w
        net.precision = self.precision
        self.net = net
        self._ws = {}

    def workspace(self, N, S, with_gradient):
        key = (N, S, bool(with_gradient))
        if key not in self._ws:
            n = _lib.lib().harp_vgg16_ws_bytes(N, S, int(with_gradient))
            if n == 0:
                raise ValueError(f"perceptual term: image size {S} is not a multiple of 8")
            self._ws[key] = torch.zeros(n, dtype=torch.uint8, device=self.dev)
        return self._ws[key]

    def features(self, image, mask, rows=None, N=None, out=None, all_slots=False):
        """activations of image[rows] * mask[rows] (image (T,S,S,3), mask (T,S,S); rows int32 (N,) or None = the first N), NHWC.
        Default: the list of the four taps (written into `out` when given).  all_slots: `out` / the result is the 13-slot list of
        activation_shapes() (None entries are not kept)."""
        S = image.shape[1]
        N = int(rows.shape[0]) if rows is not None else (image.shape[0] if N is None else N)
        if all_slots:
            if out is None:
                out = [torch.empty((N,) + s, device=self.dev) for s in activation_shapes(S)]
            slots = list(out)
        else:
            if out is None:
                out = [torch.empty((N,) + s, device=self.dev) for s in tap_shapes(S)]
            slots = [None] * 13
            for k, o in zip(TAPS, out):
                slots[k] = o
        arr = (ctypes.c_void_p * 13)(*[_lib.ptr(o) for o in slots])
        ws = self.workspace(N, S, False)
        rc = _lib.lib().harp_vgg16_features(ctypes.byref(self.net), _lib.ptr(image), _lib.ptr(mask), _lib.ptr(rows), N, S, _lib.ptr(ws), arr,
                                            _lib.stream())
        _lib.check(rc, "harp_vgg16_features")
        return out

    def term(self, rgb, y_true, mask, rows, target, target_by_row, g_rgb, loss, weight=1.0, covered=None, bound=None, side_streams=()):
        """enqueue the whole term: *loss (a float32 HIP scalar / 1-element view) = the term, g_rgb updated in place (include/harp_hip.h).
        target: the four tap tensors, or — with bound = active_tiles(mask) — the 13-slot cache of ALL activations of the target frames
        (bounded mode: the stack runs only where the rendered image can differ from its target frame).  side_streams: up to three more torch streams
        for further parts of the batch (forked and joined inside the call)"""
        N, S = rgb.shape[0], rgb.shape[1]
        t = _lib.Vgg16TermArgs()
        t.rgb, t.y_true, t.mask, t.rows = _lib.ptr(rgb), _lib.ptr(y_true), _lib.ptr(mask), _lib.ptr(rows)
        taps = [target[k] for k in TAPS] if len(target) == 13 else list(target)
        for k in range(4):
            t.target[k] = _lib.ptr(taps[k])
        if bound is not None:
            for k, slot in INPUT_SLOT.items():
                t.target_in[k] = _lib.ptr(target[slot])
            for lv, (tiles, order, count, mx, origin, pitch, side) in enumerate(bound):
                t.tiles[lv], t.tile_list[lv], t.tile_count[lv], t.max_tiles[lv] = _lib.ptr(tiles), _lib.ptr(order), _lib.ptr(count), mx
                t.tile_origin[lv], t.tile_pitch[lv], t.tile_side[lv] = _lib.ptr(origin), pitch, side
        t.target_by_row, t.covered, t.g_rgb, t.weight, t.loss = int(target_by_row), _lib.ptr(covered), _lib.ptr(g_rgb), float(weight), _lib.ptr(loss)
        t.N, t.S, t.ws = N, S, _lib.ptr(self.workspace(N, S, True))
        for i, st in enumerate(list(side_streams)[:3]):          # the batch in len + 1 parts on as many streams (include/harp_hip.h)
            t.side_streams[i] = st.cuda_stream
        _lib.check(_lib.lib().harp_vgg16_term(ctypes.byref(self.net), ctypes.byref(t), _lib.stream()), "harp_vgg16_term")


    def first_layer_gradient_filters(self):
        """data-gradient filters of the first convolution for harp_conv3x3 (64 -> 3 channels, padded to 64): the whole-term entry point has a
        vector-ALU kernel for this layer (w0t); the module-level call runs it as one more matrix-core convolution"""
        if getattr(self, "_ft0", None) is None:
            self._ft0 = conv_hip.pack_filters(self._w0, self.precision, transpose=True)
        return self._ft0


class Vgg16Rows(torch.autograd.Function):
    """`Vgg16Features.forward` on a HIP tensor (model/vgg.py:38-56 of the reference): x (N,3,H,W) -> one row per image — the flattened
    input and the flattened relu1_2 / relu2_2 / relu3_3 / relu4_3 maps (NCHW order), scaled by layers_weights — with the ten convolutions
    on csrc/conv.hip (harp_conv3x3, RELU epilogue with the fused 2x2 pool) and, backward, their data gradients (GATE / UNPOOL epilogues:
    ReLU mask and arg-max routing fused; the filters are frozen, model/vgg.py:34-36).  The reference loop body's
    `l1_loss(vgg(a), vgg(b))` (optimize_sequence.py:546-547) thereby lands on this repo's kernels, not on a library."""

    POOL_BELOW = (2, 4, 7)                    # convolutions that read the 2x2-pooled map of the layer below

    @staticmethod
    def forward(ctx, x, hip):
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] % 8 or x.shape[3] % 8:
            raise ValueError(f"Vgg16Features on HIP: input must be (N,3,H,W) with H, W multiples of 8, got {tuple(x.shape)}")
        C = conv_hip
        N, _, H, W = x.shape
        x = x.float()
        img = torch.zeros(N, H, W, 16, device=x.device)                      # NHWC, 3 -> 16 channels (the kernels walk 16 at a time)
        img[..., :3] = x.permute(0, 2, 3, 1)
        acts, h = [], img
        for k in range(10):
            d, co = _DIV[k], _COUT[k]
            out = torch.empty(N, H // d, W // d, co, device=x.device)
            pool = torch.empty(N, H // (2 * d), W // (2 * d), co, device=x.device) if (k + 1) in Vgg16Rows.POOL_BELOW else None
            f, _, b = hip.packed[k]
            C.conv3x3(h, f, co, bias=b, epilogue=C.RELU, precision=hip.precision, out=out, pooled=pool)
            acts.append(out)
            h = pool if pool is not None else out
        lw = hip.layers_weights
        row = torch.cat([lw[0] * x.flatten(1)] + [lw[i + 1] * acts[k].permute(0, 3, 1, 2).flatten(1) for i, k in enumerate(TAPS)], 1)
        ctx.hip, ctx.acts, ctx.shape = hip, acts, (N, H, W)
        return row

    @staticmethod
    def backward(ctx, g_row):
        C, hip, acts = conv_hip, ctx.hip, ctx.acts
        N, H, W = ctx.shape
        lw = hip.layers_weights
        g_row = g_row.float()
        off = 3 * H * W
        g_x = lw[0] * g_row[:, :off].reshape(N, 3, H, W)
        g_tap = {}
        for i, k in enumerate(TAPS):                                            # d / d relu(conv k), gated by the ReLU, NHWC
            n, hk, wk, ck = acts[k].shape
            seg = g_row[:, off:off + hk * wk * ck].reshape(n, ck, hk, wk).permute(0, 2, 3, 1)
            g_tap[k] = (lw[i + 1] * seg * (acts[k] > 0)).contiguous()
            off += hk * wk * ck
        G = g_tap[9]
        for k in range(9, 0, -1):
            _, ft, _ = hip.packed[k]
            below = acts[k - 1]
            if k in Vgg16Rows.POOL_BELOW:      # through the pool into the tap layer below: added to that tap's own gradient
                out = g_tap[k - 1]
                C.conv3x3(G, ft, below.shape[-1], epilogue=C.UNPOOL, precision=hip.precision, out=out, gate=below)
            else:
                out = torch.empty_like(below)
                C.conv3x3(G, ft, below.shape[-1], epilogue=C.GATE, precision=hip.precision, out=out, gate=below)
            G = out
        out = torch.empty(N, H, W, 64, device=G.device)
        C.conv3x3(G, hip.first_layer_gradient_filters(), 64, epilogue=C.GATE, precision=hip.precision, out=out, gate=torch.ones_like(out))
        return g_x + out[..., :3].permute(0, 3, 1, 2), None


__all__ = ["Vgg16Hip", "Vgg16Rows", "tap_shapes", "activation_shapes", "active_tiles", "active_share", "TILE_SIDES", "feature_length"]
