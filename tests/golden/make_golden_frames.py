"""Generates the committed input-wire-format fixture of SURVEY.md §8(f) rank 2 (tests/golden/frames/ + frames_expected.npz):

    python tests/golden/make_golden_frames.py

A 6-frame synthetic sequence in the reference's directory layout (utils/data_util.py:76-195):
    frames/img/1/unscreen_cropped/000N.jpg, frames/img/1/mask/000N_mask.jpg, frames/metro/1/metro_mano/000N_mano.pkl
and the tensors the reference's `load_multiple_sequences` -> `ImagesDataset.__getitem__` (utils/data_util.py:32-51, 76-195) produce for
them: the reference's own module is IMPORTED from /root/reference (round 4; rounds 1-3 restated `load_img`) with a stub `cv2` whose
`erode` — the one call the build image cannot make, `cv2.erode(img, np.ones((3,3), np.uint8), iterations=2)` — is written out in plain
NumPy loops from the OpenCV documentation of `erode`:
    dst(x, y) = min over (x', y') in the 3x3 neighbourhood anchored at its centre of src(x + x', y + y'), applied `iterations` times;
    default borderType = BORDER_CONSTANT with borderValue = morphologyDefaultBorderValue(), which for erosion means +DBL_MAX: pixels
    outside the image never win the minimum.  (Replicating the border pixels — BORDER_REPLICATE — gives the same result for a 3x3
    minimum: every replicated value is already inside the window; the script asserts this on every frame.)
The JPEG decode is PIL's in the reference too (`Image.open(path).convert('RGB' | 'L')`); the decoded arrays are stored so a change of
the decoder would be noticed.  The fixture is data: JPEG / pickle inputs and expected arrays, no reference source."""
import os
import pickle

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
S, N = 64, 6


def erode_3x3_loops(src, iterations, border):
    """OpenCV `erode` with a 3x3 all-ones kernel, centre anchor, written as loops.  border: "constant_max" (OpenCV's default for
    erosion) or "replicate"."""
    a = np.asarray(src, np.float64)
    H, W = a.shape
    for _ in range(iterations):
        out = np.empty_like(a)
        for y in range(H):
            for x in range(W):
                m = np.inf
                for dy in (-1, 0, 1):
                    for dx in (-1, 0, 1):
                        yy, xx = y + dy, x + dx
                        if border == "replicate":
                            yy, xx = min(max(yy, 0), H - 1), min(max(xx, 0), W - 1)
                        elif not (0 <= yy < H and 0 <= xx < W):
                            continue                      # borderValue = +DBL_MAX: never the minimum
                        m = min(m, a[yy, xx])
                out[y, x] = m
        a = out
    return a


def import_reference_data_util():
    """The reference's OWN utils/data_util.py, imported from /root/reference (build container only) with a stub `cv2` module whose `erode`
    is the documented loop above — the one call the build image cannot make.  Decode, /255 scaling, channel order, down-sampling, the
    un-thresholded mask, what is eroded and how often, the METRO pickle reading and the dataset ordering are then the reference's own
    statements (utils/data_util.py:11-51, 54-73, 76-195)."""
    import importlib.util
    import sys
    import types
    cv2 = types.ModuleType("cv2")

    def erode(src, kernel, iterations=1):
        k = np.asarray(kernel)
        assert k.shape == (3, 3) and (k == 1).all(), "the stub implements the 3x3 all-ones kernel the reference passes"
        a = np.asarray(src)
        assert a.ndim == 2 or (a.ndim == 3 and a.shape[2] == 1)
        a2 = a[..., 0] if a.ndim == 3 else a                 # cv2.erode returns (H,W) for an (H,W,1) input
        out = erode_3x3_loops(a2, iterations, "constant_max")
        assert np.array_equal(out, erode_3x3_loops(a2, iterations, "replicate"))
        return out
    cv2.erode = erode
    had = sys.modules.get("cv2")
    sys.modules["cv2"] = cv2
    try:
        spec = importlib.util.spec_from_file_location("ref_data_util", "/root/reference/utils/data_util.py")
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        if had is None:
            del sys.modules["cv2"]
        else:
            sys.modules["cv2"] = had
    return mod


def main():
    rng = np.random.default_rng(20260929)
    root = os.path.join(HERE, "frames")
    for d in ("img/1/unscreen_cropped", "img/1/mask", "metro/1/metro_mano"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    yy, xx = np.mgrid[0:S, 0:S]
    exp = {"rgb": [], "mask": [], "eroded": []}
    for i in range(N):
        name = f"{i + 1:04d}"
        # a blob that moves, touches the image border in some frames, has a hole and a one-pixel-wide spur (erosion removes it)
        cx, cy, r = 20 + 5 * i, 30 - 4 * i, 14 + i
        m = ((xx - cx) ** 2 + (yy - cy) ** 2 < r * r) & ~((xx - cx - 3) ** 2 + (yy - cy + 2) ** 2 < 9)
        m[40 + i, 5:60] = True
        if i % 2 == 0:
            m[0:4, 10:30] = True
        base = np.stack([(xx * 3 + i * 20) % 256, (yy * 2 + 40) % 256, ((xx + yy) * 2) % 256], -1).astype(np.float64)
        rgb = np.clip(base * (0.4 + 0.6 * m[..., None]) + rng.normal(0, 6, (S, S, 3)), 0, 255).astype(np.uint8)
        Image.fromarray(rgb).save(os.path.join(root, "img/1/unscreen_cropped", name + ".jpg"), quality=92)
        Image.fromarray(m.astype(np.uint8) * 255).save(os.path.join(root, "img/1/mask", name + "_mask.jpg"), quality=92)
        frame = {"joints": rng.normal(size=(1, 21, 3)).astype(np.float32) * 30, "verts": rng.normal(size=(1, 778, 3)).astype(np.float32) * 30,
                 "rot": rng.normal(size=(1, 3)).astype(np.float32) * 0.3, "pose": rng.normal(size=(1, 45)).astype(np.float32) * 0.2,
                 "shape": rng.normal(size=(1, 10)).astype(np.float32) * 0.5, "trans": np.zeros((1, 3), np.float32),
                 "cam": np.asarray((0.9 + 0.01 * i, 0.02 * i, -0.03), np.float32)}
        with open(os.path.join(root, "metro/1/metro_mano", name + "_mano.pkl"), "wb") as f:
            pickle.dump(frame, f, protocol=2)
    # ---- expected tensors: the reference's loader run on the tree just written
    R = import_reference_data_util()
    mano, ds, val_mano, val_ds = R.load_multiple_sequences(os.path.join(root, "metro") + "/", os.path.join(root, "img") + "/", train_list=["1"],
                                                           val_list=[], average_cam_sequence=False, use_smooth_seq=False, model_type="harp")
    assert len(ds) == N
    for i in range(N):
        fid, col, mask, eroded = ds[i]
        assert fid == i
        exp["rgb"].append(col.numpy()); exp["mask"].append(mask.numpy()); exp["eroded"].append(eroded.numpy())
    out = {k: np.stack(v) for k, v in exp.items()}
    out["image_paths"] = np.asarray([os.path.relpath(p, root) for p in ds.image_paths])      # dataset order = the reference's
    for k in ("pose", "rot", "trans", "shape", "cam", "joints"):
        out["mano_" + k] = np.asarray(mano[k])
    old_path = os.path.join(HERE, "frames_expected.npz")
    if os.path.exists(old_path):
        old = np.load(old_path)
        for k in ("rgb", "mask", "eroded"):
            print(f"{k}: identical to the committed fixture = {np.array_equal(old[k], out[k])}")
    np.savez_compressed(old_path, **out)
    print("wrote", root, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
