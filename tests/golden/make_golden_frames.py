"""Generates the committed input-wire-format fixture of SURVEY.md §8(f) rank 2 (tests/golden/frames/ + frames_expected.npz):

    python tests/golden/make_golden_frames.py

A 6-frame synthetic sequence in the reference's directory layout (utils/data_util.py:76-195):
    frames/img/1/unscreen_cropped/000N.jpg, frames/img/1/mask/000N_mask.jpg, frames/metro/1/metro_mano/000N_mano.pkl
and the tensors the reference's `ImagesDataset.__getitem__` (utils/data_util.py:32-51) produces for them, computed HERE by a literal
restatement of `load_img` (utils/data_util.py:11-30) in which the one call the build image cannot make, `cv2.erode(img, np.ones((3,3),
np.uint8), iterations=2)`, is written out in plain NumPy loops from the OpenCV documentation of `erode`:
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


def reference_load_img(path, load_mask=False, erode=False, downsample_factor=1):
    """utils/data_util.py:11-30, statement for statement (torch_tensor=True only wraps the array in torch.Tensor = float32)"""
    if load_mask:
        img = np.asarray(Image.open(path).convert("L")) / 255
        img = img[::downsample_factor, ::downsample_factor, None]
        if erode:
            a = erode_3x3_loops(img[..., 0], 2, "constant_max")      # cv2.erode returns (H,W) for an (H,W,1) input
            assert np.array_equal(a, erode_3x3_loops(img[..., 0], 2, "replicate"))
            img = a
    else:
        img = np.asarray(Image.open(path).convert("RGB")) / 255
        img = img[::downsample_factor, ::downsample_factor, 0:3]
    return img.astype(np.float32)


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
        img_p = os.path.join(root, "img/1/unscreen_cropped", name + ".jpg")
        msk_p = os.path.join(root, "img/1/mask", name + "_mask.jpg")
        exp["rgb"].append(reference_load_img(img_p))
        exp["mask"].append(reference_load_img(msk_p, load_mask=True))
        exp["eroded"].append(reference_load_img(msk_p, load_mask=True, erode=True))
    np.savez_compressed(os.path.join(HERE, "frames_expected.npz"), **{k: np.stack(v) for k, v in exp.items()})
    print("wrote", root, {k: np.stack(v).shape for k, v in exp.items()})


if __name__ == "__main__":
    main()
