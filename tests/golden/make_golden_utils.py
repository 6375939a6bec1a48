"""Generate golden vectors for the host-side helpers by IMPORTING THE REFERENCE in the build container.

    python tests/golden/make_golden_utils.py          # needs /root/reference (absent on the GPU box)

Writes (data only) next to this script:
  saved_params_ref.pkl   written by the REFERENCE's utils/file_utils.py:save_result (:6-17) from a seeded parameter dict
  utils_ref.npz          the arrays of that dict + what the REFERENCE's load_result / set_require_grad (:19-37) return for it
                         (keys, which ones became nn.Parameter) + PyTMinMaxScaler / scale_value / get_upscale_mano_vert_colors
                         outputs (utils/opt_utils.py:25-45)
  config_ref.json        the dict utils/config_utils.py:get_config (:5-47) returns and the config.yaml text it writes
It also checks, here, the other direction: a checkpoint written by harp_amd's save_result is read back by the REFERENCE's
load_result with identical contents ("saved_params.pkl compatibility both ways")."""
import json
import os
import pickle
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, "/root/reference")


def seeded_params(with_disps=True):
    g = torch.Generator().manual_seed(42)
    T = 5
    r = lambda *s: torch.randn(*s, generator=g)
    p = {"trans": r(T, 3) * 0.01, "pose": r(T, 45) * 0.3, "rot": r(T, 3) * 0.3, "shape": r(10) * 0.5, "wrist_pose": torch.zeros(T, 3),
         "init_joints": r(T, 21, 3) * 40.0, "verts_rgb": torch.rand(778, 3, generator=g), "verts_uvs": None, "faces_uvs": None,
         "texture": torch.rand(1, 8, 8, 3, generator=g), "uv_mask": (torch.rand(8, 8, generator=g) > 0.5).double(),
         "normal_map": torch.tensor([0.0, 0.0, 1.0]).repeat(1, 8, 8, 1), "light_positions": torch.tensor(((-0.5, -0.5, -0.5),)).repeat(T, 1),
         "amb_ratio": torch.tensor(0.4), "mesh_faces": torch.randint(0, 778, (12, 3), generator=g), "cam": r(T, 3) * 0.1 + 1.0}
    if with_disps:
        p["verts_disps"] = r(3093, 1) * 1e-3
    return p


def main():
    import utils.file_utils as RF
    import utils.config_utils as RC
    import utils.opt_utils as RO
    from harp_amd.utils import file_utils as MF

    out = {}
    # ---- file_utils: reference writes, reference reads (verts_disps left out of the read-back: set_require_grad moves it to 'cuda', :33)
    P = seeded_params()
    with tempfile.TemporaryDirectory() as d:
        RF.save_result({k: (torch.nn.Parameter(v) if (v is not None and v.is_floating_point() and k != "init_joints") else v) for k, v in P.items()}, d)
        blob = open(os.path.join(d, "saved_params.pkl"), "rb").read()
    open(os.path.join(HERE, "saved_params_ref.pkl"), "wb").write(blob)
    for k, v in P.items():
        if v is not None:
            out["p_" + k] = v.numpy()
    P2 = seeded_params(with_disps=False)
    with tempfile.TemporaryDirectory() as d:
        RF.save_result(P2, d, test=True)
        assert os.path.exists(os.path.join(d, "saved_params_test.pkl"))
        L = RF.load_result(d, device="cpu", test=True)
    out["load_keys"] = np.array(sorted(L.keys()))
    out["load_param_keys"] = np.array(sorted(k for k, v in L.items() if isinstance(v, torch.nn.Parameter)))
    out["load_none_keys"] = np.array(sorted(k for k, v in L.items() if v is None))
    for k, v in L.items():
        if v is not None:
            assert torch.equal(v.detach(), P2[k]), k
    # ---- the other direction: harp_amd writes, the REFERENCE reads
    with tempfile.TemporaryDirectory() as d:
        MF.save_result(P2, d)
        L2 = RF.load_result(d, device="cpu")
        raw = pickle.load(open(os.path.join(d, "saved_params.pkl"), "rb"))
    assert sorted(L2.keys()) == sorted(L.keys())
    for k, v in L2.items():
        assert (v is None and P2[k] is None) or torch.equal(v.detach(), P2[k]), k
        assert isinstance(v, torch.nn.Parameter) == isinstance(L[k], torch.nn.Parameter), k
    assert all(v is None or isinstance(v, np.ndarray) for v in raw.values())
    print("mirror-written checkpoint read back by the reference's load_result: identical")

    # ---- config_utils
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as d:
        os.chdir(d)
        try:
            cfg = RC.get_config()
            yaml_text = open(os.path.join(cfg["base_output_dir"], "config.yaml")).read()
        finally:
            os.chdir(cwd)
    json.dump({"config": cfg, "yaml": yaml_text}, open(os.path.join(HERE, "config_ref.json"), "w"), indent=1, sort_keys=True)

    # ---- opt_utils
    g = torch.Generator().manual_seed(7)
    x = torch.randn(3, 50, generator=g)
    x[1] = 2.5                                             # a constant channel: dist == 0 -> 1 (:37)
    out["scale_in"] = x.numpy().copy()
    out["scale_out"] = RO.scale_value(x.clone()).numpy()
    out["scaler_out"] = RO.PyTMinMaxScaler()(x.clone()).numpy()
    v = torch.randn(200, 3, generator=g).numpy() * np.array([30.0, 5.0, 80.0]) + 3.0
    out["upscale_in"] = v
    out["upscale_out"] = RO.get_upscale_mano_vert_colors(v)
    np.savez_compressed(os.path.join(HERE, "utils_ref.npz"), **out)
    print("written: saved_params_ref.pkl utils_ref.npz config_ref.json")


if __name__ == "__main__":
    main()
