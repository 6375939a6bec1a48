"""Input wire format of a HARP fitting job: mirror of the reference's `utils/data_util.py` API.

Directory layout (utils/data_util.py:76-195, produced by metro_modifications/end2end_inference_handmesh.py:250-265):

    <image_dir>/<seq>/unscreen_cropped/<name>.jpg       RGB frame
    <image_dir>/<seq>/mask/<name>_mask.jpg              hand mask
    <metro_output_dir>/<seq>/metro_mano/<name>_mano.pkl {'joints' (1,21,3) mm, 'verts', 'rot' (1,3), 'pose' (1,45), 'shape' (1,10),
                                                         'trans' (1,3), 'cam' (3,), ...}

Same names, arguments and return values as the reference (`load_img`, `ImagesDataset`, `combine_dict_to_batch`,
`load_multiple_sequences`, `load_sample_sequence`).  Differences: no cv2 (absent here) — the 3x3 erosion x2 of the mask
(utils/data_util.py:17-20) is a numpy minimum filter with cv2.erode's default border rule (neighbours outside the image are ignored);
and `ResidentTargets`, which decodes a dataset once into the three tensors `FitEngine.set_targets` keeps in HBM (the reference
re-decodes every frame every epoch in 20 DataLoader workers, optimize_sequence.py:399, 446-450).
"""
import os
import pickle

import numpy as np
import torch
from PIL import Image
from torch.utils.data import Dataset


def _erode3x3(img, iterations=2):
    """cv2.erode(img, np.ones((3,3)), iterations=n) for a 2-D float array: minimum over the 3x3 neighbourhood, out-of-image
    neighbours ignored (cv2's default border value for erosion is +inf)."""
    a = np.asarray(img, dtype=np.float64)
    for _ in range(iterations):
        p = np.pad(a, 1, mode="constant", constant_values=np.inf)
        a = np.minimum.reduce([p[dy:dy + a.shape[0], dx:dx + a.shape[1]] for dy in range(3) for dx in range(3)])
    return a


def load_img(img_path, torch_tensor=False, downsample_factor=1, load_mask=False, erode=False):
    """utils/data_util.py:11-30.  RGB: (H,W,3) in [0,1].  Mask: (H,W,1) in [0,1]; with erode=True the result is 2-D (H,W), exactly
    like cv2.erode drops the singleton channel in the reference."""
    if load_mask:
        img = np.asarray(Image.open(img_path).convert("L")) / 255
        img = img[::downsample_factor, ::downsample_factor, None]
        if erode:
            img = _erode3x3(img[..., 0], iterations=2)
    else:
        img = np.asarray(Image.open(img_path).convert("RGB")) / 255
        img = img[::downsample_factor, ::downsample_factor, 0:3]
    if torch_tensor:
        img = torch.Tensor(img)
    return img


class ImagesDataset(Dataset):
    """utils/data_util.py:32-51: item i -> (fid, rgb (H,W,3), mask (H,W,1), eroded mask (H,W)) as float32 tensors."""

    def __init__(self, images_paths, mask_paths, downsample_factor):
        self.image_paths = images_paths
        self.mask_paths = mask_paths
        self.downsample_factor = downsample_factor

    def __len__(self):
        return len(self.image_paths)

    def __getitem__(self, ix):
        fid = ix
        col_img = load_img(self.image_paths[fid], downsample_factor=self.downsample_factor, torch_tensor=True)
        mask_img = load_img(self.mask_paths[fid], downsample_factor=self.downsample_factor, torch_tensor=True, load_mask=True)
        mask_img_eroded = load_img(self.mask_paths[fid], downsample_factor=self.downsample_factor, torch_tensor=True, load_mask=True,
                                   erode=True)
        return fid, col_img, mask_img, mask_img_eroded


def combine_dict_to_batch(mano_dict):
    """utils/data_util.py:54-73: list of per-frame dicts -> dict of stacked tensors ('cam' as is, 'seq' stays a list, everything else
    loses its leading singleton axis)."""
    keys = list(mano_dict[0].keys())
    out = {k: [] for k in keys}
    for frame in mano_dict:
        for k in keys:
            if k == "cam":
                out[k].append(torch.from_numpy(np.asarray(frame[k])))
            elif k == "seq":
                out[k].append(frame[k])
            else:
                out[k].append(torch.from_numpy(np.asarray(frame[k]).squeeze(0)))
    for k in keys:
        if k != "seq":
            out[k] = torch.stack(out[k])
    return out


def _read_frame(mano_filename, seq, cam_list):
    with open(mano_filename, "rb") as f:
        mano_param = pickle.load(f)
    mano_param["seq"] = seq
    cam_list.setdefault(seq, []).append(mano_param["cam"])
    return mano_param


def _average_cams(cam_list, *param_lists):
    """"Force the same camera for the entire sequence" (utils/data_util.py:170-182)."""
    avg = {seq: np.mean(cams, axis=0) for seq, cams in cam_list.items()}
    for params in param_lists:
        for mano_param in params:
            mano_param["cam"] = avg[mano_param["seq"]]


def _pkl_names(folder):
    return sorted(fn[:-9] for fn in os.listdir(folder) if fn.endswith(".pkl"))          # "0001_mano.pkl" -> "0001"


def load_multiple_sequences(metro_output_dir, image_dir, max_size=0, val=False, val_size=0, average_cam_sequence=False,
                            train_list=("1", "2", "3", "4", "5"), val_list=("6", "7", "8", "9"), use_smooth_seq=False, model_type="harp"):
    """utils/data_util.py:76-195 -> (mano_params, images_dataset, val_mano_params, val_images_dataset).  Frames are ordered by
    (sequence name, frame name); an empty val_list makes the validation set the training set."""
    pkl_folder = "metro_mano_smooth" if use_smooth_seq else "metro_mano"
    if model_type == "nimble":
        pkl_folder = "nimble_" + pkl_folder

    def collect(seqs):
        names = sorted((seq, n) for seq in seqs for n in _pkl_names(os.path.join(metro_output_dir, seq, pkl_folder)))
        return names

    cam_list = {}

    def read(names):
        imgs, masks, manos = [], [], []
        for seq, name in names:
            imgs.append(os.path.join(image_dir, seq, "unscreen_cropped", name + ".jpg"))
            masks.append(os.path.join(image_dir, seq, "mask", name + "_mask.jpg"))
            manos.append(_read_frame(os.path.join(metro_output_dir, seq, pkl_folder, name + "_mano.pkl"), seq, cam_list))
        return imgs, masks, manos

    image_paths, mask_paths, mano_list = read(collect(train_list))
    val_names = collect(val_list) if len(val_list) > 0 else []
    val_image_paths, val_mask_paths, val_mano_list = read(val_names)
    if average_cam_sequence:
        _average_cams(cam_list, mano_list, val_mano_list)
    if len(val_names) == 0:
        val_image_paths, val_mask_paths, val_mano_list = image_paths, mask_paths, mano_list
    return (combine_dict_to_batch(mano_list), ImagesDataset(image_paths, mask_paths, downsample_factor=1),
            combine_dict_to_batch(val_mano_list), ImagesDataset(val_image_paths, val_mask_paths, downsample_factor=1))


def load_sample_sequence(metro_output_dir, image_dir, max_size=0, val=False, val_size=0, average_cam_sequence=False):
    """utils/data_util.py:196-285: a flat directory of `<name>_mano.pkl`; images `<image_dir><name>.jpg` for captured videos
    ("sequence" / "interhand" in the path) else `<name>_cropped.jpg`; masks `<name>_mask.jpg`.  val=True: 90/10 split (or max_size
    frames for training and the rest / val_size for validation)."""
    names = _pkl_names(metro_output_dir)
    if val:
        if max_size == 0:
            max_size, val_size = (len(names) * 9) // 10, len(names) // 10
        elif val_size == 0:
            val_size = len(names) - max_size
    if max_size == 0:
        max_size = len(names)
    captured = ("sequence" in metro_output_dir) or ("interhand" in metro_output_dir)
    cam_list = {}
    train, valid = ([], [], []), ([], [], [])
    for name in names:
        img = image_dir + name + (".jpg" if captured else "_cropped.jpg")
        mask = image_dir + name + "_mask.jpg"
        mano_param = _read_frame(metro_output_dir + name + "_mano.pkl", "0", cam_list)
        if len(train[0]) < max_size:
            dst = train
        elif len(valid[0]) < val_size:
            dst = valid
        else:
            break
        dst[0].append(img); dst[1].append(mask); dst[2].append(mano_param)
    if average_cam_sequence:
        _average_cams(cam_list, train[2], valid[2])
    if val_size == 0:
        valid = train
    return (combine_dict_to_batch(train[2]), ImagesDataset(train[0], train[1], downsample_factor=1),
            combine_dict_to_batch(valid[2]), ImagesDataset(valid[0], valid[1], downsample_factor=1))


class ResidentTargets:
    """Decode an `ImagesDataset` ONCE into the tensors the fitting engine keeps resident in HBM:
    y_true (T,S,S,3), y_sil (T,S,S), y_sil_col (T,S,S) — `FitEngine.set_targets(*ResidentTargets(ds, frames).tensors())`.
    `frames` selects / orders the items (e.g. one rank's shard, harp_amd.dist.shard_frames)."""

    def __init__(self, dataset, frames=None, device="cpu", pin=False):
        idx = range(len(dataset)) if frames is None else list(frames)
        items = [dataset[i] for i in idx]
        self.fid = torch.tensor([int(it[0]) for it in items], dtype=torch.int32)
        self.y_true = torch.stack([torch.as_tensor(it[1], dtype=torch.float32) for it in items])
        S0, S1 = self.y_true.shape[1:3]
        self.y_sil = torch.stack([torch.as_tensor(it[2], dtype=torch.float32).reshape(S0, S1) for it in items])
        self.y_sil_col = torch.stack([torch.as_tensor(it[3], dtype=torch.float32).reshape(S0, S1) for it in items])
        if pin and device != "cpu":
            self.y_true, self.y_sil, self.y_sil_col = (t.pin_memory() for t in (self.y_true, self.y_sil, self.y_sil_col))
        if str(device) != "cpu":
            self.y_true, self.y_sil, self.y_sil_col = (t.to(device, non_blocking=pin) for t in (self.y_true, self.y_sil, self.y_sil_col))

    def tensors(self):
        return self.y_true, self.y_sil, self.y_sil_col

    def __len__(self):
        return self.y_true.shape[0]
