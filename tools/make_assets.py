"""Build the topology/UV asset files under harp_amd/assets/ from the reference's template data.

Run ONCE in the build container (reads /root/reference/template, which does not exist on the GPU
box).  Outputs are DATA (topology, UV coordinates, masks, a procedurally deformed base geometry),
not source.  Also re-verifies SURVEY.md §4 item 1: the template OBJs reproduce the SubdivideMeshes
face/corner order exactly under one injective vertex relabelling.

    python tools/make_assets.py
"""
import os
import pickle
import sys

import numpy as np
from PIL import Image

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "harp_amd", "assets")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from harp_amd.topology import subdivide_topology  # noqa: E402


def load_obj(path):
    v, vt, fv, ft = [], [], [], []
    for line in open(path):
        p = line.split()
        if not p:
            continue
        if p[0] == "v":
            v.append([float(x) for x in p[1:4]])
        elif p[0] == "vt":
            vt.append([float(x) for x in p[1:3]])
        elif p[0] == "f":
            a = [q.split("/") for q in p[1:4]]
            fv.append([int(q[0]) - 1 for q in a])
            ft.append([int(q[1]) - 1 for q in a])
    return np.array(v, np.float64), np.array(vt, np.float64), np.array(fv, np.int64), np.array(ft, np.int64)


def relabel(faces_sub, faces_obj, n_sub):
    """injective map subdivided-index -> OBJ index implied by equal face/corner order."""
    m = -np.ones(n_sub, np.int64)
    for a, b in zip(faces_sub.reshape(-1), faces_obj.reshape(-1)):
        if m[a] == -1:
            m[a] = b
        elif m[a] != b:
            raise SystemExit("KAT FAILED: conflicting relabelling")
    assert (m >= 0).all() and len(np.unique(m)) == n_sub, "KAT FAILED: not injective"
    return m


def deform(v):
    """Deterministic smooth warp so the committed geometry is not the licensed template itself."""
    c = v.mean(0)
    x = (v - c)
    s = np.abs(x).max()
    u = x / s
    w = np.stack([
        0.035 * np.sin(2.1 * u[:, 1] + 0.3) + 0.02 * np.cos(3.0 * u[:, 2]),
        0.030 * np.sin(1.7 * u[:, 2] - 0.5) + 0.02 * np.cos(2.6 * u[:, 0]),
        0.040 * np.sin(2.4 * u[:, 0] + 0.9) + 0.02 * np.cos(2.2 * u[:, 1]),
    ], 1)
    return (u * np.array([1.04, 0.97, 1.08]) + w) * s


def build(name, faces0, obj_path, mask_path, unit_scale):
    v_obj, vt, f_obj, ft = load_obj(obj_path)
    V0 = int(faces0.max()) + 1
    edges0, faces_sub = subdivide_topology(faces0, V0)
    assert faces_sub.shape == f_obj.shape, (faces_sub.shape, f_obj.shape)
    m = relabel(faces_sub, f_obj, V0 + len(edges0))
    verts_sub = v_obj[m] * unit_scale                      # metres, subdivided-index order
    # KAT part 2: OBJ midpoints really are midpoints of the relabelled originals (loose: template was posed)
    mid = verts_sub[:V0][edges0].mean(1)
    err = np.abs(mid - verts_sub[V0:]).max()
    print(f"[{name}] V0={V0} E0={len(edges0)} V={V0+len(edges0)} F={len(faces_sub)} VT={len(vt)} "
          f"midpoint max err {err:.2e} m")
    base = deform(verts_sub[:V0]).astype(np.float32)
    mask = np.asarray(Image.open(mask_path).convert("L").resize((512, 512)))  # optimize_sequence.py:174-178
    np.savez_compressed(os.path.join(OUT, f"{name}_template.npz"),
                        faces0=faces0.astype(np.int32), verts_uvs=vt.astype(np.float32),
                        faces_uvs=ft.astype(np.int32), base_verts=base, uv_mask=mask.astype(np.uint8))


def main():
    os.makedirs(OUT, exist_ok=True)
    corr = pickle.load(open(f"{REF}/template/arm/smplx_arm_corr.pkl", "rb"), encoding="latin1")
    build("hand", np.asarray(corr["mano_face"]), f"{REF}/template/hand/textured_hand.obj",
          f"{REF}/template/hand/uv_mask.png", 1e-3)   # hand OBJ is in mm
    build("arm", np.asarray(corr["face"]), f"{REF}/template/arm/arm_template.obj",
          f"{REF}/template/arm/uv_mask.png", 1.0)     # arm OBJ is in m
    np.savez_compressed(os.path.join(OUT, "arm_corr.npz"),
                        mano_vert_from_arm=np.asarray(corr["mano_vert_from_arm"], np.int32),
                        arm_joint=np.asarray(corr["arm_joint"], np.int32),
                        mano_joint=np.asarray(corr["mano_joint"], np.int32))


if __name__ == "__main__":
    main()
