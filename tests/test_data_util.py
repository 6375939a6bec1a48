"""CPU tests of the input wire format mirror (harp_amd/utils/data_util.py; reference utils/data_util.py — cv2 is absent here, so the
reference module itself cannot be imported; the erosion is checked against scipy's minimum filter with cv2.erode's border rule)."""
import os
import pickle

import numpy as np
import pytest
import torch
from PIL import Image

from harp_amd.utils import data_util as D


def _frame(rng, cam):
    return {"joints": rng.normal(size=(1, 21, 3)).astype(np.float32), "verts": rng.normal(size=(1, 778, 3)).astype(np.float32),
            "rot": rng.normal(size=(1, 3)).astype(np.float32), "pose": rng.normal(size=(1, 45)).astype(np.float32),
            "shape": rng.normal(size=(1, 10)).astype(np.float32), "trans": np.zeros((1, 3), np.float32), "cam": np.asarray(cam, np.float32)}


@pytest.fixture()
def tree(tmp_path):
    rng = np.random.default_rng(0)
    S = 32
    for seq, names in (("1", ["0002", "0001", "0010"]), ("2", ["0001"]), ("6", ["0003", "0001"])):
        os.makedirs(tmp_path / "metro" / seq / "metro_mano")
        os.makedirs(tmp_path / "img" / seq / "unscreen_cropped")
        os.makedirs(tmp_path / "img" / seq / "mask")
        for i, n in enumerate(names):
            with open(tmp_path / "metro" / seq / "metro_mano" / f"{n}_mano.pkl", "wb") as f:
                pickle.dump(_frame(rng, (0.9 + 0.01 * i, 0.1 * int(seq), -0.05)), f)
            Image.fromarray(rng.integers(0, 255, (S, S, 3), dtype=np.uint8)).save(tmp_path / "img" / seq / "unscreen_cropped" / f"{n}.jpg")
            m = np.zeros((S, S), np.uint8)
            m[6:26, 4:20] = 255
            m[0:3, 0:9] = 255                                   # touches the border: cv2.erode ignores out-of-image neighbours
            Image.fromarray(m).save(tmp_path / "img" / seq / "mask" / f"{n}_mask.jpg")
    return tmp_path


def test_load_multiple_sequences_layout_order_and_shapes(tree):
    mp, ds, vmp, vds = D.load_multiple_sequences(str(tree / "metro"), str(tree / "img"), train_list=["1", "2"], val_list=["6"])
    assert len(ds) == 4 and len(vds) == 2
    assert mp["seq"] == ["1", "1", "1", "2"]                     # sorted by (sequence, frame name)
    assert [os.path.basename(p) for p in ds.image_paths] == ["0001.jpg", "0002.jpg", "0010.jpg", "0001.jpg"]
    assert mp["pose"].shape == (4, 45) and mp["joints"].shape == (4, 21, 3) and mp["cam"].shape == (4, 3) and mp["shape"].shape == (4, 10)
    fid, rgb, mask, er = ds[2]
    assert fid == 2 and rgb.shape == (32, 32, 3) and mask.shape == (32, 32, 1) and er.shape == (32, 32)
    assert rgb.dtype == torch.float32 and 0.0 <= float(rgb.min()) and float(rgb.max()) <= 1.0
    # an empty val_list makes validation = training
    mp2, ds2, vmp2, vds2 = D.load_multiple_sequences(str(tree / "metro"), str(tree / "img"), train_list=["1"], val_list=[])
    assert vds2.image_paths == ds2.image_paths and torch.equal(vmp2["pose"], mp2["pose"])


def test_average_cam_per_sequence(tree):
    mp, *_ = D.load_multiple_sequences(str(tree / "metro"), str(tree / "img"), train_list=["1", "2"], val_list=["6"], average_cam_sequence=True)
    cams = mp["cam"].numpy()
    assert np.allclose(cams[0], cams[1]) and np.allclose(cams[1], cams[2]) and not np.allclose(cams[2], cams[3])
    assert np.allclose(cams[0], [0.91, 0.1, -0.05], atol=1e-6)     # mean of 0.90, 0.91, 0.92


def test_mask_erosion_matches_cv2_rule(tree):
    from scipy import ndimage
    path = str(tree / "img" / "1" / "mask" / "0001_mask.jpg")
    m = D.load_img(path, load_mask=True)[..., 0]
    want = m
    for _ in range(2):
        want = ndimage.minimum_filter(want, size=3, mode="constant", cval=np.inf)
    got = D.load_img(path, load_mask=True, erode=True)
    assert got.shape == m.shape and np.array_equal(got, want)
    assert got[0, 0] > 0.5 and got[2, 5] < 0.5 and got[10, 10] > 0.5 and got[7, 5] < 0.5       # border kept, 2-px rim removed


def test_load_sample_sequence_split_and_resident_targets(tmp_path):
    rng = np.random.default_rng(1)
    d = tmp_path / "sequence_x"
    os.makedirs(d)
    for i in range(10):
        n = f"{i:04d}"
        with open(d / f"{n}_mano.pkl", "wb") as f:
            pickle.dump(_frame(rng, (1.0, 0.0, 0.0)), f)
        Image.fromarray(rng.integers(0, 255, (16, 16, 3), dtype=np.uint8)).save(d / f"{n}.jpg")
        Image.fromarray((rng.random((16, 16)) > 0.5).astype(np.uint8) * 255).save(d / f"{n}_mask.jpg")
    mp, ds, vmp, vds = D.load_sample_sequence(str(d) + os.sep, str(d) + os.sep, val=True)
    assert len(ds) == 9 and len(vds) == 1 and mp["pose"].shape == (9, 45)
    rt = D.ResidentTargets(ds, frames=[3, 1])
    y_true, y_sil, y_col = rt.tensors()
    assert y_true.shape == (2, 16, 16, 3) and y_sil.shape == (2, 16, 16) and y_col.shape == (2, 16, 16) and rt.fid.tolist() == [3, 1]
    assert torch.equal(y_true[1], ds[1][1]) and torch.equal(y_sil[0], ds[3][2][..., 0]) and (y_col <= y_sil + 1e-6).all()


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_committed_frames_fixture_pins_load_img_and_erosion():
    """tests/golden/frames (6 JPEG frames + masks + METRO pickles in the reference's layout) against frames_expected.npz, the output of
    the REFERENCE's own utils/data_util.py (load_multiple_sequences -> ImagesDataset) imported by tests/golden/make_golden_frames.py
    with a stub cv2 whose erode is written out as loops from the OpenCV documentation: decode, /255 scaling, channel order,
    un-thresholded mask, 3x3 erosion x2, shapes and dtypes, dataset order and the stacked METRO parameters."""
    exp = np.load(os.path.join(GOLDEN, "frames_expected.npz"))
    root = os.path.join(GOLDEN, "frames")
    mp, ds, _, _ = D.load_multiple_sequences(os.path.join(root, "metro"), os.path.join(root, "img"), train_list=["1"], val_list=[])
    assert len(ds) == 6 and mp["pose"].shape == (6, 45) and mp["cam"].shape == (6, 3)
    assert [os.path.basename(p) for p in ds.image_paths] == [f"{i:04d}.jpg" for i in range(1, 7)]
    assert [os.path.relpath(p, root) for p in ds.image_paths] == exp["image_paths"].tolist()            # the reference's dataset order
    for k in ("pose", "rot", "trans", "shape", "cam", "joints"):                                         # combine_dict_to_batch, :54-73
        assert mp[k].dtype == torch.float32 and np.array_equal(mp[k].numpy(), exp["mano_" + k]), k
    for i in range(6):
        fid, rgb, mask, er = ds[i]
        assert fid == i and rgb.dtype == mask.dtype == er.dtype == torch.float32
        assert np.array_equal(rgb.numpy(), exp["rgb"][i]) and np.array_equal(mask.numpy(), exp["mask"][i])
        assert np.array_equal(er.numpy(), exp["eroded"][i]), i
    # the masks are JPEGs: values are NOT thresholded by the reference (utils/data_util.py:14) and neither here
    frac = ((exp["mask"] > 0.02) & (exp["mask"] < 0.98)).mean()
    assert 0.0 < frac < 0.2, frac
    assert (exp["eroded"] <= exp["mask"][..., 0] + 1e-7).all() and exp["eroded"].sum() < 0.8 * exp["mask"].sum()
    # frames 1, 3, 5 touch the top border: erosion keeps the border rows (out-of-image neighbours never win the minimum)
    assert exp["eroded"][0][0, 15] > 0.9 and exp["eroded"][0][3, 15] < 0.5 + 0.5 * exp["eroded"][0][2, 15]
    rt = D.ResidentTargets(ds)
    assert np.array_equal(rt.y_true.numpy(), exp["rgb"]) and np.array_equal(rt.y_sil.numpy(), exp["mask"][..., 0])
    assert np.array_equal(rt.y_sil_col.numpy(), exp["eroded"]) and rt.fid.tolist() == list(range(6))


def test_load_obj_uvs_reads_what_pytorch3d_load_obj_returns(tmp_path):
    """utils/hand_model_utils.load_obj_uvs: `vt` lines -> verts_uvs, second index of every face corner -> faces.textures_idx
    (what utils/hand_model_utils.py:58-60 takes from pytorch3d.io.load_obj); the template assets were built from the same fields"""
    from harp_amd.utils.hand_model_utils import load_obj_uvs
    p = tmp_path / "t.obj"
    p.write_text("# tiny\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nvt 0.0 0.0\nvt 1.0 0.0\nvt 0.0 1.0\nvt 1.0 1.0\nvt 0.5 0.5\n"
                 "f 1/1 2/2 3/3\nf 2/2/1 4/5/1 3/3/1\n")
    uv, f = load_obj_uvs(str(p))
    assert uv.shape == (5, 2) and uv.dtype == torch.float32 and f.dtype == torch.int64
    assert f.tolist() == [[0, 1, 2], [1, 4, 2]] and uv[4].tolist() == [0.5, 0.5]
    (tmp_path / "q.obj").write_text("v 0 0 0\nvt 0 0\nf 1/1 1/1 1/1 1/1\n")
    import pytest
    with pytest.raises(ValueError):
        load_obj_uvs(str(tmp_path / "q.obj"))
    # relative (negative) texture indices count back from the END of the vt list (as load_obj resolves them: also when vt and f lines
    # interleave); corners without a texture index give -1
    (tmp_path / "i.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvt 1 0\nf 1/-3 2/-2 3/-1\nvt 0 1\n")
    assert load_obj_uvs(str(tmp_path / "i.obj"))[1].tolist() == [[0, 1, 2]]
    (tmp_path / "r.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 0 1\nf 1/-3 2/-2 3/-1\nf 1//1 2//1 3//1\nf 1 2 3\n")
    uv, f = load_obj_uvs(str(tmp_path / "r.obj"))
    assert f.tolist() == [[0, 1, 2], [-1, -1, -1], [-1, -1, -1]]
    (tmp_path / "s.obj").write_text("v 0 0 0\nvt 0 0\nf 1/2 1/1 1/1\n")
    with pytest.raises(ValueError, match="out of range"):
        load_obj_uvs(str(tmp_path / "s.obj"))
