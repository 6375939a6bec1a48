"""CPU: the VGG16 feature module of the perceptual term (harp_amd/model/vgg.py) against the oracle's functional restatement with
the same filters; state-dict layouts; loud failure without weights.  (Pretrained filters do not exist here: parity unpinned.)"""
import pytest
import torch

from harp_amd.model.vgg import Vgg16Features, feature_length

LW = [1, 1 / 16, 1 / 8, 1 / 4, 1]          # optimize_sequence.py:405


def _filters(vgg):
    sd = vgg.state_dict()
    return {int(k.split(".")[1]): (sd[k], sd[k.replace("weight", "bias")]) for k in sd if k.endswith("weight")}


def test_layout_matches_vgg16_features():
    vgg = Vgg16Features(layers_weights=LW, weights="random")
    shapes = {k: tuple(v.shape) for k, v in vgg.state_dict().items() if k.endswith("weight")}
    assert shapes == {"slice1.0.weight": (64, 3, 3, 3), "slice1.2.weight": (64, 64, 3, 3), "slice2.5.weight": (128, 64, 3, 3),
                      "slice2.7.weight": (128, 128, 3, 3), "slice3.10.weight": (256, 128, 3, 3), "slice3.12.weight": (256, 256, 3, 3),
                      "slice3.14.weight": (256, 256, 3, 3), "slice4.17.weight": (512, 256, 3, 3), "slice4.19.weight": (512, 512, 3, 3),
                      "slice4.21.weight": (512, 512, 3, 3)}
    assert not any(p.requires_grad for p in vgg.parameters())
    assert Vgg16Features(weights="random").layers_weights == [1 / 32, 1 / 16, 1 / 8, 1 / 4, 1]       # model/vgg.py:16-17


def test_forward_against_oracle_and_l1_of_parts():
    from oracle import harp_ref as H
    vgg = Vgg16Features(layers_weights=LW, weights="random", seed=3)
    torch.manual_seed(0)
    x, y = torch.rand(2, 3, 32, 40), torch.rand(2, 3, 32, 40)
    out = vgg(x)
    assert out.shape == (2, feature_length(32, 40))
    ref = H.vgg16_features(_filters(vgg), x, LW)
    assert torch.allclose(out, ref, rtol=1e-5, atol=1e-6)
    # the engine sums |a-b| slice by slice instead of concatenating: same number
    fa, fb = vgg.features(x), vgg.features(y)
    parts = sum((a - b).abs().sum() for a, b in zip(fa, fb)) / (2 * out.shape[1])
    assert abs(parts.item() - torch.nn.functional.l1_loss(vgg(x), vgg(y)).item()) < 1e-6
    # gradient w.r.t. the image flows although the filters are frozen
    xg = x.clone().requires_grad_(True)
    torch.nn.functional.l1_loss(vgg(xg), vgg(y)).backward()
    assert xg.grad.abs().sum() > 0


def test_state_dict_layouts(tmp_path):
    src = Vgg16Features(layers_weights=LW, weights="random", seed=1)
    x = torch.rand(1, 3, 16, 16)
    # torchvision's layout ("features.N.*", plus entries this module does not use) from a file
    tv = {f"features.{k.split('.')[1]}.{k.split('.')[2]}": v for k, v in src.state_dict().items()}
    tv["features.24.weight"] = torch.zeros(512, 512, 3, 3)
    tv["classifier.0.weight"] = torch.zeros(4, 4)
    path = tmp_path / "vgg16.pth"
    torch.save(tv, path)
    assert torch.equal(Vgg16Features(layers_weights=LW, weights=str(path))(x), src(x))
    # the reference module's own layout
    assert torch.equal(Vgg16Features(layers_weights=LW, weights=src.state_dict())(x), src(x))
    del tv["features.21.bias"]
    with pytest.raises(KeyError):
        Vgg16Features(weights=tv)


def test_no_silent_random_fallback():
    try:
        import torchvision  # noqa: F401
        pytest.skip("torchvision present: weights=None takes the pretrained filters")
    except ImportError:
        pass
    with pytest.raises(RuntimeError):
        Vgg16Features()


def test_active_tiles_cover_every_feature_difference():
    """the premise of the perceptual term's bounded mode (harp_amd/model/vgg_hip.active_tiles, csrc/conv.hip): vgg(a * mask) and vgg(b * mask)
    can differ only inside the tiles (16 or 8 pixels a side per level) the mask's support reaches through the receptive field — at EVERY one of the ten convolutions, not
    only at the four taps (checked with the torch module on the CPU; levels = image side S, S/2, S/4, S/8)"""
    from harp_amd.model.vgg import Vgg16Features
    from harp_amd.model.vgg_hip import active_tiles
    S = 96
    g = torch.Generator().manual_seed(0)
    vgg = Vgg16Features(weights="random", seed=4).double()
    mask = torch.zeros(2, S, S, dtype=torch.float64)
    mask[0, 40:43, 50:52] = 1.0                       # a few pixels
    mask[1, 0:20, 70:96] = (torch.rand(20, 26, generator=g) > 0.5).double()
    a, b = torch.rand(2, S, S, 3, generator=g, dtype=torch.float64), torch.rand(2, S, S, 3, generator=g, dtype=torch.float64)
    xa, xb = (a * mask[..., None]).permute(0, 3, 1, 2), (b * mask[..., None]).permute(0, 3, 1, 2)
    for sides in ((16, 8, 8, 8), (16, 16, 16, 16), (16, 16, 8, 8)):
        bound = active_tiles(mask, tile_sides=sides)
        level, ha, hb, checked = 0, xa, xb, 0
        with torch.no_grad():
            for n in range(1, 5):
                for layer in getattr(vgg, f"slice{n}"):
                    ha, hb = layer(ha), layer(hb)
                    if isinstance(layer, torch.nn.MaxPool2d):
                        level += 1
                    if not isinstance(layer, torch.nn.ReLU):
                        continue
                    tiles, order, count, mx, origin, G, side = bound[level]
                    assert side == sides[level]
                    differs = (ha != hb).any(1)                                    # (2, s, s)
                    assert differs.any()
                    for f in range(2):
                        ys, xs = torch.nonzero(differs[f], as_tuple=True)
                        cell = ((ys + origin[f, 0]) // side) * G + (xs + origin[f, 1]) // side      # tile (ty, tx) covers [side ty - oy, +side) x [side tx - ox, +side)
                        assert tiles[f][cell].all(), (n, level, f)
                        assert origin[f, 0] % 2 == 0 and origin[f, 1] % 2 == 0 and 0 <= origin[f].min() and origin[f].max() <= side - 2
                        # the lists hold exactly the flagged tiles, in raster order
                        assert order[f, :count[f]].tolist() == torch.nonzero(tiles[f]).flatten().tolist()
                    checked += 1
        assert checked == 10 and bound[0][2][0] <= 4                               # (frame 0: a few pixels -> at most 2 x 2 level-0 tiles)
        # the shifted grid never needs more tiles than the aligned one
        for (a, b) in zip(bound, active_tiles(mask, shift_grid=False, tile_sides=sides)):
            assert (a[2] <= b[2]).all() and (b[4] == 0).all()


def test_module_reproduces_the_reference_modules_golden_rows(golden_dir):
    """tests/golden/vgg_ref.npz was written by the REFERENCE's own model/vgg.py (imported from /root/reference by
    tests/golden/make_golden_vgg.py, with only the absent `torchvision.models.vgg16` constructor stubbed by the published layer sequence +
    seeded filters): this package's Vgg16Features, fed the same filters in torchvision's state-dict layout, returns the same rows — slices,
    taps, default and explicit layers_weights, concatenation order, state-dict keys."""
    import os
    import sys
    import numpy as np
    from harp_amd.model.vgg import Vgg16Features
    sys.path.insert(0, golden_dir)
    from vgg_filters import state_dict_torchvision_layout
    ref = np.load(os.path.join(golden_dir, "vgg_ref.npz"))
    x = torch.from_numpy(ref["x"])
    sd = state_dict_torchvision_layout()
    with torch.no_grad():
        for key, lw in (("y_default", None), ("y_fit", list(ref["layers_weights_fit"]))):
            m = Vgg16Features(layers_weights=lw, weights=sd)
            y = m(x)
            assert y.shape == ref[key].shape
            assert torch.allclose(y, torch.from_numpy(ref[key]), rtol=0, atol=2e-6), (key, (y - torch.from_numpy(ref[key])).abs().max())
            if lw is None:
                assert np.allclose(m.layers_weights, ref["layers_weights_default"])
    assert sorted(m.state_dict().keys()) == list(ref["state_dict_keys"])
    assert all(not p.requires_grad for p in m.parameters())
