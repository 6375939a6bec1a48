"""What a hierarchical (4x4-block corner) coverage test could skip in the camera-view face scan (csrc/raster_body.h): statistics of the
4x4 blocks the scan visits for the bench scene's faces (CPU, numpy; the same bbox / block walk as the kernel: blocks start at the bbox
origin clipped to the 16x16 tile).  python tools/dev/raster_block_stats.py [S]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from harp_amd import synth
from oracle import harp_ref as H, p3d_like as P3

S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
tpl = synth.load_template("hand")
topo_np = synth.build_topology(tpl["faces0"], 778)
model_np = synth.make_mano_model(tpl, seed=0)
seq, focal = synth.make_sequence(model_np, 4, S, seed=0)
model = {k: torch.as_tensor(v) for k, v in model_np.items()}
topo = {k: torch.as_tensor(v) for k, v in topo_np.items() if not np.isscalar(v)}
tot = dict(faces=0, face_tiles=0, blocks=0, empty=0, full=0, inside_px=0, bbox_px=0)
for f in range(4):
    verts_mm, _ = H.mano_forward({k: v.double() if v.is_floating_point() else v for k, v in model.items()}, torch.cat([seq["rot"][f:f+1], seq["pose"][f:f+1]], 1).double(),
                                 seq["shape"][f:f+1].double(), seq["trans"][f:f+1].double())
    v0 = (verts_mm / 1000.0)[0].numpy()
    # subdivide: midpoints appended (no displacement: statistics only)
    e = topo_np["edges0"] if "edges0" in topo_np else None
    faces = topo_np["faces"]
    if e is not None:
        v = np.concatenate([v0, 0.5 * (v0[e[:, 0]] + v0[e[:, 1]])])
    else:
        v = v0
    R, T = H.camera_RT(seq["cam"][f:f+1].double(), S, focal)
    ndc = P3.world_to_ndc(torch.as_tensor(v)[None], R, T, focal, (S / 2.0, S / 2.0), S)[1][0].numpy()
    px = (1.0 - ndc[:, 0]) * S / 2 - 0.5          # pixel coordinate of an NDC x (pixel 0 at +1 - 1/S)
    py = (1.0 - ndc[:, 1]) * S / 2 - 0.5
    for tri in faces:
        x, y = px[tri], py[tri]
        area = (x[1] - x[0]) * (y[2] - y[0]) - (y[1] - y[0]) * (x[2] - x[0])
        if area == 0:
            continue
        X0, X1 = int(np.ceil(x.min())), int(np.floor(x.max()))
        Y0, Y1 = int(np.ceil(y.min())), int(np.floor(y.max()))
        if X0 > X1 or Y0 > Y1 or X1 < 0 or Y1 < 0 or X0 >= S or Y0 >= S:
            continue
        tot["faces"] += 1
        for ty in range(max(Y0, 0) // 16, min(Y1, S - 1) // 16 + 1):
            for tx in range(max(X0, 0) // 16, min(X1, S - 1) // 16 + 1):
                x0, x1 = max(X0, tx * 16), min(X1, tx * 16 + 15, S - 1)
                y0, y1 = max(Y0, ty * 16), min(Y1, ty * 16 + 15, S - 1)
                tot["face_tiles"] += 1
                for by in range(y0, y1 + 1, 4):
                    for bx in range(x0, x1 + 1, 4):
                        xs, ys = np.meshgrid(np.arange(bx, min(bx + 3, x1) + 1), np.arange(by, min(by + 3, y1) + 1))
                        s = -np.sign(area)        # (edge functions below are edge(p; a, b) = (px-ax)(by-ay) - (py-ay)(bx-ax); `area` above has the opposite sign)
                        e0 = ((xs - x[1]) * (y[2] - y[1]) - (ys - y[1]) * (x[2] - x[1])) * s
                        e1 = ((xs - x[2]) * (y[0] - y[2]) - (ys - y[2]) * (x[0] - x[2])) * s
                        e2 = ((xs - x[0]) * (y[1] - y[0]) - (ys - y[0]) * (x[1] - x[0])) * s
                        ins = (e0 > 0) & (e1 > 0) & (e2 > 0)
                        tot["blocks"] += 1
                        tot["empty"] += int(ins.sum() == 0)
                        tot["full"] += int(ins.sum() == 16)
                        tot["inside_px"] += int(ins.sum())
                        tot["bbox_px"] += ins.size
b = tot["blocks"]
print(f"S = {S}: {tot['faces']} faces on screen over 4 frames, {tot['face_tiles']} (face, tile) pairs, {b} 4x4 blocks visited = {b / tot['face_tiles']:.2f} per pair")
print(f"  blocks without an inside pixel (what a trivial-reject corner test could skip): {100 * tot['empty'] / b:.1f} %")
print(f"  blocks fully inside (what a trivial accept could shorten): {100 * tot['full'] / b:.1f} %")
print(f"  lanes of 16 inside the bbox: {tot['bbox_px'] / b:.1f}; inside the face: {tot['inside_px'] / b:.1f}")
