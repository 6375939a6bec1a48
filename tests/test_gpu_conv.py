"""GPU: the matrix-core 3x3 convolution of the perceptual term (csrc/conv.hip through the C ABI) against torch's float64 convolution and
its autograd on the CPU — forward with fused bias / ReLU / max pool / L1 tap, and the two data-gradient epilogues (ReLU gate, max-pool
routing).  Both arithmetic modes: float32 MFMA (bound 5e-6 of the output scale: a float32 fma chain in another order) and the three-term bf16
split (bound 3e-5: 2^-16 per product)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = {0: 5e-6, 1: 3e-5}


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def _nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def _case(Cin, Cout, H, W, N=2, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, Cin, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(Cout, Cin, 3, 3, generator=g, dtype=torch.float64) * (2.0 / (9 * Cin)) ** 0.5
    b = torch.randn(Cout, generator=g, dtype=torch.float64) * 0.1
    return x, w, b


def _err(got, want):
    return ((got.double().cpu() - want).abs().max() / want.abs().max()).item()


SHAPES = [(16, 64, 32, 32), (64, 64, 40, 24), (64, 128, 32, 32), (128, 256, 16, 16), (256, 512, 16, 16), (512, 512, 8, 8)]


@pytest.mark.parametrize("precision", [0, 1])
@pytest.mark.parametrize("shape", SHAPES)
def test_forward_relu_pool(shape, precision):
    from harp_amd.model import conv_hip as C
    Cin, Cout, H, W = shape
    x, w, b = _case(*shape)
    want = F.relu(F.conv2d(x, w, b, padding=1))
    xd, wd, bd = _nhwc(x).float().to(DEV), w.float().to(DEV), b.float().to(DEV)
    filt = C.pack_filters(wd, precision)
    out = torch.full((x.shape[0], H, W, Cout), float("nan"), device=DEV)
    pooled = torch.full((x.shape[0], H // 2, W // 2, Cout), float("nan"), device=DEV)
    C.conv3x3(xd, filt, Cout, bias=bd, epilogue=C.RELU, precision=precision, out=out, pooled=pooled)
    torch.cuda.synchronize()
    assert _err(_nchw(out), want) < TOL[precision]
    assert _err(_nchw(pooled), F.max_pool2d(want, 2, 2)) < TOL[precision]
    # pooled only (the full-size activation is not written)
    pooled2 = torch.empty_like(pooled)
    C.conv3x3(xd, filt, Cout, bias=bd, epilogue=C.RELU, precision=precision, out=None, pooled=pooled2)
    assert torch.equal(pooled2, pooled)


@pytest.mark.parametrize("precision", [0, 1])
def test_tap_epilogue_loss_and_gradient(precision):
    from harp_amd.model import conv_hip as C
    Cin, Cout, H, W = 64, 128, 32, 32
    x, w, b = _case(Cin, Cout, H, W, N=3)
    g = torch.Generator().manual_seed(9)
    T = 5
    target = torch.relu(torch.randn(T, Cout, H, W, generator=g, dtype=torch.float64))
    rows = torch.tensor([4, 0, 2])
    conv = F.conv2d(x, w, b, padding=1).requires_grad_(True)
    act = F.relu(conv)
    scale = 0.37
    loss = scale * (act - target[rows]).abs().sum()
    (g_conv,) = torch.autograd.grad(loss, conv)
    xd = _nhwc(x).float().to(DEV)
    filt = C.pack_filters(w.float().to(DEV), precision)
    out = torch.empty(3, H, W, Cout, device=DEV)
    pooled = torch.empty(3, H // 2, W // 2, Cout, device=DEV)
    g_tap = torch.empty_like(out)
    acc = torch.zeros(1, dtype=torch.float64, device=DEV)
    C.conv3x3(xd, filt, Cout, bias=b.float().to(DEV), epilogue=C.RELU_TAP, precision=precision, out=out, pooled=pooled,
              target=_nhwc(target).float().to(DEV), target_row=rows.int().to(DEV), tap_scale=scale, g_tap=g_tap, loss=acc)
    torch.cuda.synchronize()
    assert _err(_nchw(out), act.detach()) < TOL[precision]
    assert abs(acc.item() - loss.item()) < 10 * TOL[precision] * loss.item()
    # the sign of a feature difference at rounding level is not decided: compare where |difference| is clear of the arithmetic's error
    diff = (act.detach() - target[rows]).abs()
    unclear = ((diff < 1e-3) & (diff > 0)) | (conv.detach().abs() < 1e-3)
    d = (_nchw(g_tap).double().cpu() - g_conv)[~unclear]
    assert unclear.float().mean() < 0.01 and d.abs().max().item() < 1e-6 * scale


@pytest.mark.parametrize("precision", [0, 1])
@pytest.mark.parametrize("shape", [(64, 64, 40, 24), (128, 64, 32, 32), (512, 256, 16, 16)])
def test_data_gradient_through_relu(shape, precision):
    """d/d(conv_a) of conv_b(relu(conv_a)): g_a = conv(g_b, w_b^T mirrored) * [relu(conv_a) > 0]"""
    from harp_amd.model import conv_hip as C
    Cout_b, Cin_b, H, W = shape            # conv_b: Cin_b -> Cout_b; the gradient convolution runs Cout_b -> Cin_b
    g = torch.Generator().manual_seed(3)
    a = torch.randn(2, Cin_b, H, W, generator=g, dtype=torch.float64).requires_grad_(True)
    w = torch.randn(Cout_b, Cin_b, 3, 3, generator=g, dtype=torch.float64) * (2.0 / (9 * Cin_b)) ** 0.5
    g_b = torch.randn(2, Cout_b, H, W, generator=g, dtype=torch.float64)
    act = F.relu(a)
    (want,) = torch.autograd.grad(F.conv2d(act, w, None, padding=1), a, g_b)
    filt = C.pack_filters(w.float().to(DEV), precision, transpose=True)
    out = torch.full((2, H, W, Cin_b), float("nan"), device=DEV)
    C.conv3x3(_nhwc(g_b).float().to(DEV), filt, Cin_b, epilogue=C.GATE, precision=precision, out=out, gate=_nhwc(act.detach()).float().to(DEV))
    torch.cuda.synchronize()
    assert _err(_nchw(out), want) < TOL[precision]


@pytest.mark.parametrize("precision", [0, 1])
def test_data_gradient_through_max_pool(precision):
    """tap layer -> pool -> conv_b: the gradient arriving at the tap's convolution = its own L1 gradient + conv_b's data gradient routed
    through max_pool2d's arg-max and the ReLU (ties between equal maxima: the first in row-major order, as torch)"""
    from harp_amd.model import conv_hip as C
    Cin_b, Cout_b, H, W = 64, 128, 16, 24          # pooled size; the tap activation is (2H, 2W)
    g = torch.Generator().manual_seed(4)
    pre = torch.randn(2, Cin_b, 2 * H, 2 * W, generator=g, dtype=torch.float64)
    pre = (pre * 2).round() / 2                      # many exact ties and many zeros: the routing rules are exercised
    pre.requires_grad_(True)
    w = torch.randn(Cout_b, Cin_b, 3, 3, generator=g, dtype=torch.float64) * (2.0 / (9 * Cin_b)) ** 0.5
    g_b = torch.randn(2, Cout_b, H, W, generator=g, dtype=torch.float64)
    g_own = torch.randn(2, Cin_b, 2 * H, 2 * W, generator=g, dtype=torch.float64)
    act = F.relu(pre)
    (want,) = torch.autograd.grad(F.conv2d(F.max_pool2d(act, 2, 2), w, None, padding=1), pre, g_b)
    want = want + g_own
    filt = C.pack_filters(w.float().to(DEV), precision, transpose=True)
    out = _nhwc(g_own).float().to(DEV)
    C.conv3x3(_nhwc(g_b).float().to(DEV), filt, Cin_b, epilogue=C.UNPOOL, precision=precision, out=out, gate=_nhwc(act.detach()).float().to(DEV))
    torch.cuda.synchronize()
    assert _err(_nchw(out), want) < TOL[precision]


def test_bad_arguments_are_refused():
    from harp_amd import _lib
    from harp_amd.model import conv_hip as C
    x = torch.zeros(1, 8, 8, 24, device=DEV)         # 24 input channels: not a multiple of 16
    f = torch.zeros(_lib.lib().harp_conv3x3_filter_bytes(64, 24), dtype=torch.uint8, device=DEV)
    with pytest.raises(RuntimeError, match="harp_conv3x3"):
        C.conv3x3(x, f, 64, out=torch.zeros(1, 8, 8, 64, device=DEV))


@pytest.mark.parametrize("precision", [0, 1])
def test_whole_term_against_torch_autograd(precision):
    """harp_vgg16_features / harp_vgg16_term against the torch module (harp_amd/model/vgg.py == reference model/vgg.py) in float64 on the
    CPU: tap features, the term's value and d term / d rgb, with cached (all target frames, indexed by row) and per-step target features"""
    from harp_amd.model.vgg import Vgg16Features
    from harp_amd.model.vgg_hip import Vgg16Hip
    S, N, T = 64, 2, 3
    LW = [1, 1 / 16, 1 / 8, 1 / 4, 1]
    vgg = Vgg16Features(layers_weights=LW, weights="random", seed=1)
    g = torch.Generator().manual_seed(11)
    rgb = torch.rand(N, S, S, 3, generator=g, dtype=torch.float64).requires_grad_(True)
    y_true = torch.rand(T, S, S, 3, generator=g, dtype=torch.float64)
    mask = (torch.rand(T, S, S, generator=g, dtype=torch.float64) > 0.3).double()
    mask[:, ::7] *= 0.5                                              # (fractional values: eroded / resized masks)
    rows = torch.tensor([2, 0])
    vgg64 = Vgg16Features(layers_weights=LW, weights=vgg.state_dict()).double()
    m = mask[rows].unsqueeze(-1)
    fp = vgg64((rgb * m).permute(0, 3, 1, 2))
    ft = vgg64((y_true[rows] * m).permute(0, 3, 1, 2))
    want = F.l1_loss(fp, ft)
    (g_want,) = torch.autograd.grad(want, rgb)

    hip = Vgg16Hip(vgg, DEV, precision)
    rgb_d, yt_d, mask_d, rows_d = (t.detach().float().to(DEV).contiguous() for t in (rgb, y_true, mask, rows))
    rows_d = rows.int().to(DEV)
    feats = hip.features(yt_d, mask_d)                               # all T frames: the cache of the four taps
    ref_feats = vgg64.features((y_true * mask.unsqueeze(-1)).permute(0, 3, 1, 2), skip_input=True, weighted=False)
    for f, r, shp in zip(feats, ref_feats, ((64, S, S), (128, S // 2, S // 2), (256, S // 4, S // 4), (512, S // 8, S // 8))):
        assert _err(_nchw(f), r.detach().reshape(T, *shp)) < 4 * TOL[precision]
    covered = torch.randint(-1, 5, (N, S, S), generator=g).int()
    g0 = torch.randn(N, S, S, 3, generator=g)
    expect = torch.where((covered >= 0).unsqueeze(-1), g0.double() + 0.7 * g_want, torch.zeros((), dtype=torch.float64))
    for by_row in (1, 0):
        target = feats if by_row else hip.features(yt_d, mask_d, rows_d)
        g_rgb = g0.to(DEV).contiguous()
        loss = torch.zeros(1, device=DEV)
        hip.term(rgb_d, yt_d, mask_d, rows_d, target, by_row, g_rgb, loss, weight=0.7, covered=covered.to(DEV))
        torch.cuda.synchronize()
        assert abs(loss.item() - want.item()) < 2e-5 * want.item(), (loss.item(), want.item())
        rel = ((g_rgb.double().cpu() - expect).norm() / expect.norm()).item()
        print(f"[whole VGG term, precision {precision}, cached={by_row}] loss {loss.item():.6f} vs {want.item():.6f}, gradient rel-L2 {rel:.1e}")
        assert rel < (2e-3, 6e-2)[precision], rel


@pytest.mark.parametrize("precision", [0, 1])
def test_bounded_term_equals_the_full_pass(precision):
    """harp_vgg16_term's bounded mode (the stack runs only in the tiles — 16 pixels a side at S, 8 below by default — the mask's support reaches through the receptive field, and
    reads the cached TARGET activations next to them) against the full pass on the same inputs: same loss, same gradient — and both against
    torch's float64 autograd.  Masks with a small support (few active tiles at the three finer levels), a support touching the image border,
    and an empty mask."""
    from harp_amd.model.vgg import Vgg16Features
    from harp_amd.model.vgg_hip import Vgg16Hip, active_tiles
    S, N, T = 128, 3, 4
    LW = [1, 1 / 16, 1 / 8, 1 / 4, 1]
    vgg = Vgg16Features(layers_weights=LW, weights="random", seed=3)
    g = torch.Generator().manual_seed(21)
    rgb = torch.rand(N, S, S, 3, generator=g, dtype=torch.float64).requires_grad_(True)
    y_true = torch.rand(T, S, S, 3, generator=g, dtype=torch.float64)
    mask = torch.zeros(T, S, S, dtype=torch.float64)
    mask[0, 70:100, 20:55] = 1.0                                     # a blob in the interior
    mask[1, 0:40, 90:128] = (torch.rand(40, 38, generator=g) > 0.2).double()    # ragged, touching two borders
    mask[3, 60:62, 60:62] = 0.5                                      # 4 pixels
    rows = torch.tensor([1, 0, 3])                                   # (frame 2: empty mask, not in this batch)
    vgg64 = Vgg16Features(layers_weights=LW, weights=vgg.state_dict()).double()
    m = mask[rows].unsqueeze(-1)
    want = F.l1_loss(vgg64((rgb * m).permute(0, 3, 1, 2)), vgg64((y_true[rows] * m).permute(0, 3, 1, 2)))
    (g_want,) = torch.autograd.grad(want, rgb)

    hip = Vgg16Hip(vgg, DEV, precision)
    rgb_d, yt_d, mask_d = (t.detach().float().to(DEV).contiguous() for t in (rgb, y_true, mask))
    rows_d = rows.int().to(DEV)
    cache = hip.features(yt_d, mask_d, all_slots=True)
    bound = active_tiles(mask_d)
    counts = [b[2].tolist() for b in bound]
    assert counts[0][2] == 0 and counts[0][0] < 16 and counts[0][1] < 16 and counts[0][3] == 1, counts     # frame 2 holds no tile at all
    res = {}
    for name, bd in (("full", None), ("bounded", bound)):
        g_rgb = torch.zeros(N, S, S, 3, device=DEV)
        loss = torch.zeros(1, device=DEV)
        hip.term(rgb_d, yt_d, mask_d, rows_d, cache, 1, g_rgb, loss, weight=1.0, bound=bd)
        torch.cuda.synchronize()
        res[name] = (loss.item(), g_rgb.double().cpu())
        rel = ((res[name][1] - g_want).norm() / g_want.norm()).item()
        print(f"[VGG term {name}, precision {precision}] loss {res[name][0]:.7f} vs {want.item():.7f}, gradient rel-L2 vs float64 torch {rel:.1e}")
        # (bf16 split: features are good to 1e-5 of their scale, and an L1-of-features gradient IS the signs of the differences: the 2-4 of
        #  ~1e5 non-zero differences per tap that are below 1e-5 of the activation flip, each moving the gradient by 2 / sqrt(1e5) — measured
        #  3e-3 ... 3e-2 per tap row against 3e-7 ... 1e-6 for the float32 mode, tools/dev/vgg_diag.py; the loss itself agrees to 1e-6)
        assert abs(res[name][0] - want.item()) < 2e-5 * want.item() and rel < (2e-3, 6e-2)[precision]
    assert abs(res["full"][0] - res["bounded"][0]) < 1e-6 * res["full"][0]
    assert torch.equal(res["full"][1], res["bounded"][1])            # tile by tile the same arithmetic on the same inputs
    # the other mixes of 16- and 8-pixel tile grids over the levels (the default is 16, 8, 8, 8;  8: four independent tiles per workgroup, one per wave; validity cells of 4, 8 and
    # 16 input pixels; the un-pool routing into either kind of grid): the same bits as the full pass
    for sides in ((16, 16, 16, 16), (16, 16, 8, 8), (16, 16, 16, 8)):
        for shift in (True, False):
            bd = active_tiles(mask_d, shift_grid=shift, tile_sides=sides)
            g_rgb = torch.zeros(N, S, S, 3, device=DEV)
            loss = torch.zeros(1, device=DEV)
            hip.term(rgb_d, yt_d, mask_d, rows_d, cache, 1, g_rgb, loss, weight=1.0, bound=bd)
            torch.cuda.synchronize()
            assert torch.equal(g_rgb.double().cpu(), res["full"][1]) and abs(loss.item() - res["full"][0]) < 1e-6 * res["full"][0], (sides, shift, loss.item())
    # the batch in parts on 2 - 4 streams (harp_vgg16_term_args.side_streams: uneven parts of the 3 images, more streams than images): the
    # same gradient bit for bit, the same loss up to the order its double accumulator is added in; twice in a row (the accumulator comes back zero)
    streams = [torch.cuda.Stream(device=DEV) for _ in range(3)]
    for bd in (None, bound):
        for k in (1, 2, 3, 1):
            g_rgb = torch.zeros(N, S, S, 3, device=DEV)
            loss = torch.zeros(1, device=DEV)
            hip.term(rgb_d, yt_d, mask_d, rows_d, cache, 1, g_rgb, loss, weight=1.0, bound=bd, side_streams=streams[:k])
            torch.cuda.synchronize()
            assert torch.equal(g_rgb.double().cpu(), res["full"][1]) and abs(loss.item() - res["full"][0]) < 1e-6 * res["full"][0], (bd is None, k, loss.item())
    # an all-empty batch: nothing to compute, zero loss and gradient
    g_rgb = torch.ones(1, S, S, 3, device=DEV)
    loss = torch.ones(1, device=DEV)
    hip.term(rgb_d[:1], yt_d, mask_d, torch.tensor([2], dtype=torch.int32, device=DEV), cache, 1, g_rgb, loss, weight=1.0, bound=bound)
    torch.cuda.synchronize()
    assert loss.item() == 0.0 and (g_rgb == 1.0).all()


def test_filters_from_a_torchvision_state_dict_file(tmp_path):
    """`--vgg-weights <file>` / configs["vgg_weights"] end to end: a state dict in torchvision's vgg16 layout ("features.N.weight", plus
    the classifier entries the term does not use) on disk -> Vgg16Features(weights=path) -> packed for the matrix cores -> the HIP term.
    HARP_VGG16_WEIGHTS=<vgg16-397923af.pth> runs the same check on the real pretrained filters when the file exists (there is no
    network in the build image, so by default the file is a seeded stand-in of the same layout)."""
    import os
    from harp_amd.model.vgg import Vgg16Features
    from harp_amd.model.vgg_hip import Vgg16Hip
    path = os.environ.get("HARP_VGG16_WEIGHTS")
    if not (path and os.path.exists(path)):
        src = Vgg16Features(weights="random", seed=7)
        tv = {f"features.{k.split('.')[1]}.{k.split('.')[2]}": v for k, v in src.state_dict().items()}
        tv["features.24.weight"], tv["classifier.0.weight"] = torch.zeros(512, 512, 3, 3), torch.zeros(8, 8)
        path = str(tmp_path / "vgg16_layout.pth")
        torch.save(tv, path)
    vgg = Vgg16Features(layers_weights=[1, 1 / 16, 1 / 8, 1 / 4, 1], weights=path)
    S, N = 64, 2
    g = torch.Generator().manual_seed(2)
    rgb = torch.rand(N, S, S, 3, generator=g, dtype=torch.float64).requires_grad_(True)
    y_true = torch.rand(N, S, S, 3, generator=g, dtype=torch.float64)
    mask = (torch.rand(N, S, S, generator=g) > 0.4).double()
    vgg64 = Vgg16Features(layers_weights=vgg.layers_weights, weights=vgg.state_dict()).double()
    m = mask.unsqueeze(-1)
    want = F.l1_loss(vgg64((rgb * m).permute(0, 3, 1, 2)), vgg64((y_true * m).permute(0, 3, 1, 2)))
    (g_want,) = torch.autograd.grad(want, rgb)
    hip = Vgg16Hip(vgg, DEV, 0)
    rgb_d, yt_d, mask_d = (t.detach().float().to(DEV).contiguous() for t in (rgb, y_true, mask))
    rows = torch.arange(N, dtype=torch.int32, device=DEV)
    g_rgb, loss = torch.zeros(N, S, S, 3, device=DEV), torch.zeros(1, device=DEV)
    hip.term(rgb_d, yt_d, mask_d, rows, hip.features(yt_d, mask_d), 1, g_rgb, loss)
    torch.cuda.synchronize()
    rel = ((g_rgb.double().cpu() - g_want).norm() / g_want.norm()).item()
    assert abs(loss.item() - want.item()) < 2e-5 * want.item() and rel < 2e-3, (loss.item(), want.item(), rel)


@pytest.mark.parametrize("precision", [0, 1])
def test_hip_features_reproduce_the_reference_modules_golden_rows(golden_dir, precision):
    """the HIP convolution stack against rows written by the REFERENCE's own model/vgg.py (tests/golden/vgg_ref.npz, see
    tests/golden/make_golden_vgg.py): the four taps harp_vgg16_features returns are the reference row's segments (flattened NCHW, scaled by
    layers_weights), and harp_vgg16_term's loss between two images is torch.nn.L1Loss of the reference's two rows (optimize_sequence.py:546-547)"""
    import os
    import sys
    import numpy as np
    from harp_amd.model.vgg import Vgg16Features
    from harp_amd.model.vgg_hip import Vgg16Hip
    sys.path.insert(0, golden_dir)
    from vgg_filters import state_dict_torchvision_layout
    ref = np.load(os.path.join(golden_dir, "vgg_ref.npz"))
    x = torch.from_numpy(ref["x"])                                   # (2,3,16,16)
    N, S = x.shape[0], x.shape[-1]
    LW = [float(v) for v in ref["layers_weights_fit"]]
    vgg = Vgg16Features(layers_weights=LW, weights=state_dict_torchvision_layout())
    hip = Vgg16Hip(vgg, DEV, precision)
    img = x.permute(0, 2, 3, 1).contiguous().to(DEV)                 # NHWC
    ones = torch.ones(N, S, S, device=DEV)
    taps = hip.features(img, ones)
    row = torch.from_numpy(ref["y_fit"]).double()
    off = 3 * S * S
    assert torch.allclose(row[:, :off], LW[0] * x.flatten(1).double())
    for f, w in zip(taps, LW[1:]):
        seg = w * _nchw(f).flatten(1).double().cpu()
        want = row[:, off:off + seg.shape[1]]
        assert ((seg - want).abs().max() / want.abs().max()).item() < 4 * TOL[precision]
        off += seg.shape[1]
    assert off == row.shape[1]
    # the term: image 0 against image 1 as its target (mask of ones)
    want = (row[0] - row[1]).abs().mean().item()
    g_rgb, loss = torch.zeros(1, S, S, 3, device=DEV), torch.zeros(1, device=DEV)
    rows = torch.tensor([1], dtype=torch.int32, device=DEV)
    hip.term(img[:1].contiguous(), img, ones, rows, taps, 1, g_rgb, loss)
    torch.cuda.synchronize()
    assert abs(loss.item() - want) < 2e-5 * want, (loss.item(), want)


def _ref_filters_module(golden_dir, LW):
    import sys
    from harp_amd.model.vgg import Vgg16Features
    sys.path.insert(0, golden_dir)
    from vgg_filters import state_dict_torchvision_layout
    return Vgg16Features(layers_weights=LW, weights=state_dict_torchvision_layout())


@pytest.mark.parametrize("precision", [0, 1])
def test_module_call_on_hip_reproduces_the_reference_rows_64x48(golden_dir, precision):
    """`Vgg16Features.forward` on a HIP tensor (model/vgg_hip.py:Vgg16Rows, the ten convolutions on csrc/conv.hip) against rows the
    REFERENCE's module wrote for a 64 x 48 input (tests/golden/make_golden_vgg.py): non-square, 4 x 3 tiles of 16 x 16 at full resolution,
    seams between tiles at every level, one ragged 8 x 6 tile at the last.  The fixture keeps every 5th element of each row, the absolute
    sum of each of its five segments and the L1 distance of the two rows; and `l1_loss(vgg(a), vgg(b))` (optimize_sequence.py:546-547,
    verbatim) gives the reference's number."""
    import os
    import numpy as np
    ref = np.load(os.path.join(golden_dir, "vgg_ref.npz"))
    LW = [float(v) for v in ref["layers_weights_fit"]]
    vgg = _ref_filters_module(golden_dir, LW)
    vgg.hip_precision = precision
    x = torch.from_numpy(ref["x_64x48"]).to(DEV)
    with torch.no_grad():
        row = vgg(x)
    torch.cuda.synchronize()
    seg = [int(v) for v in ref["y_64x48_segments"]]
    assert row.shape == (2, seg[-1])
    got, want = row[:, ::5].double().cpu(), torch.from_numpy(ref["y_64x48_every5th"]).double()
    idx = torch.arange(0, seg[-1], 5)
    for i in range(5):
        sel = (idx >= seg[i]) & (idx < seg[i + 1])
        scale = want[:, sel].abs().max()
        assert ((got[:, sel] - want[:, sel]).abs().max() / scale).item() < 4 * TOL[precision], i
        sums = row[:, seg[i]:seg[i + 1]].double().abs().sum(1).cpu()
        assert torch.allclose(sums, torch.from_numpy(ref["y_64x48_segment_abs_sums"][:, i]), rtol=2e-5), i
    l1_loss = torch.nn.L1Loss()
    with torch.no_grad():
        loss = l1_loss(vgg(x[:1]), vgg(x[1:]))
    assert abs(loss.item() - float(ref["y_64x48_l1"])) < 2e-5 * float(ref["y_64x48_l1"])


@pytest.mark.parametrize("precision", [0, 1])
def test_module_call_backward_on_hip_against_torch_float64(golden_dir, precision):
    """autograd through `Vgg16Features.forward` on HIP tensors (GATE / UNPOOL data-gradient convolutions of csrc/conv.hip) against torch's
    float64 autograd through the same module on the CPU: a LINEAR functional of the row (no sign noise), 64 x 48, then the loop's own
    `l1_loss(vgg(pred * mask), vgg(true * mask))` on the reference-written masked pair of tests/golden/vgg_ref.npz — loss and gradient
    against the numbers the imported reference module and its autograd produced."""
    import os
    import numpy as np
    from harp_amd.model.vgg import Vgg16Features
    ref = np.load(os.path.join(golden_dir, "vgg_ref.npz"))
    LW = [float(v) for v in ref["layers_weights_fit"]]
    vgg = _ref_filters_module(golden_dir, LW)
    vgg.hip_precision = precision
    vgg64 = Vgg16Features(layers_weights=LW, weights=vgg.state_dict()).double()
    g = torch.Generator().manual_seed(9)
    x = torch.from_numpy(ref["x_64x48"]).double().requires_grad_(True)
    R = torch.randn(2, vgg64(x.detach()).shape[1], generator=g, dtype=torch.float64)
    (want,) = torch.autograd.grad((vgg64(x) * R).sum(), x)
    xd = x.detach().float().to(DEV).requires_grad_(True)
    (got,) = torch.autograd.grad((vgg(xd) * R.float().to(DEV)).sum(), xd)
    torch.cuda.synchronize()
    rel = ((got.double().cpu() - want).norm() / want.norm()).item()
    # (bf16 split: activations within 1e-5 of zero change side, and with them the ReLU gates of the data gradient — 6e-3 measured, the same
    #  effect as in the whole-term tests; the float32 mode is the parity anchor)
    assert rel < (2e-6, 2e-2)[precision], rel
    # the loop body's two lines on the masked pair (reference numbers)
    l1_loss = torch.nn.L1Loss()
    y_pred = torch.from_numpy(ref["pair_pred"]).to(DEV).requires_grad_(True)
    y_true = torch.from_numpy(ref["pair_true"]).to(DEV)
    y_sil_true_col = torch.from_numpy(ref["pair_mask"]).to(DEV)[None]
    loss = l1_loss(vgg((y_pred * y_sil_true_col.unsqueeze(-1)).permute(0, 3, 1, 2)),
                   vgg((y_true * y_sil_true_col.unsqueeze(-1)).permute(0, 3, 1, 2)))
    loss.backward()
    torch.cuda.synchronize()
    g_want = torch.from_numpy(ref["pair_grad"]).double()
    rel = ((y_pred.grad.double().cpu() - g_want).norm() / g_want.norm()).item()
    print(f"[module call, precision {precision}] loss {loss.item():.8f} vs reference {float(ref['pair_loss']):.8f}, gradient rel-L2 {rel:.1e}")
    assert abs(loss.item() - float(ref["pair_loss"])) < 2e-5 * float(ref["pair_loss"]) and rel < (2e-3, 6e-2)[precision]


@pytest.mark.parametrize("shift", [True, False])
@pytest.mark.parametrize("precision", [0, 1])
def test_whole_term_full_and_bounded_against_the_reference_pair(golden_dir, precision, shift):
    """harp_vgg16_term — full pass, bounded mode, bounded mode with the shifted tile grids — on the masked 64 x 64 pair whose loss and
    image gradient the imported reference module (model/vgg.py) and torch autograd wrote into tests/golden/vgg_ref.npz."""
    import os
    import numpy as np
    from harp_amd.model.vgg_hip import Vgg16Hip, active_tiles
    ref = np.load(os.path.join(golden_dir, "vgg_ref.npz"))
    LW = [float(v) for v in ref["layers_weights_fit"]]
    hip = Vgg16Hip(_ref_filters_module(golden_dir, LW), DEV, precision)
    rgb, y_true, mask = (torch.from_numpy(ref[k]).to(DEV).contiguous() for k in ("pair_pred", "pair_true", "pair_mask"))
    mask = mask[None].contiguous()
    rows = torch.zeros(1, dtype=torch.int32, device=DEV)
    cache = hip.features(y_true, mask, all_slots=True)
    bound = active_tiles(mask, shift_grid=shift)
    assert bound[0][2].item() < 16                                   # the blob leaves full-resolution tiles out
    g_want = torch.from_numpy(ref["pair_grad"]).double()
    for bd in (None, bound):
        g_rgb, loss = torch.zeros_like(rgb), torch.zeros(1, device=DEV)
        hip.term(rgb, y_true, mask, rows, cache, 1, g_rgb, loss, weight=1.0, bound=bd)
        torch.cuda.synchronize()
        rel = ((g_rgb.double().cpu() - g_want).norm() / g_want.norm()).item()
        assert abs(loss.item() - float(ref["pair_loss"])) < 2e-5 * float(ref["pair_loss"]) and rel < (2e-3, 6e-2)[precision], (bd is None, loss.item(), rel)
