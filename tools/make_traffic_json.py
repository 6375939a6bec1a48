"""profiles/traffic_latest.json from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) over the same bench command:
    python tools/make_traffic_json.py FETCH.db WRITE.db "<command the passes profiled>" > profiles/traffic_latest.json
Per-launch HBM bytes of bench.py's kernel groups = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 (KiB counters; gfx950 FETCH_SIZE
under-reports coalesced reads 2x - calibrated on the Adam kernel)."""
import json
import sqlite3
import sys
from collections import defaultdict


LAST = 4      # launches per kernel that belong to the profiled steps (bench.py --steps 3 --warmup 1): bench.py first renders its targets
              # with the same kernels in other modes (dense outputs, gradient image instead of the fused loss) — those launches are left out


def per_launch(path, counter, last=LAST):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(pmc_events)")]
    idx = {c: i for i, c in enumerate(cols)}
    per = defaultdict(lambda: defaultdict(float))
    for r in cur.execute("select * from pmc_events"):
        name = r[idx["name"]] if "name" in idx else r[idx["kernel_name"]]
        if r[idx["counter_name"]] != counter:
            continue
        per[name][r[idx["dispatch_id"]]] += r[idx["value"] if "value" in idx else idx["counter_value"]]
    out = {}
    for k, d in per.items():
        ids = sorted(d)
        # set-up kernels run once per view: keep the launches of the last `last` steps of both views
        n = last * (2 if any(t in k for t in ("face_setup_kernel", "bin_faces_kernel", "expand_bits_kernel", "order_tiles_kernel")) else 1)
        keep = ids[-n:]
        out[k] = (sum(d[i] for i in keep) / len(keep), len(keep))
    return out


# group -> parts; a part = tuple of ALTERNATIVE substrings of the kernel name (at least one must match a profiled kernel, otherwise the
# script fails: in round 2 the rasteriser kernels gained a second template argument, "raster_kernel<1>" silently stopped matching and
# the committed traffic file lost its rasteriser counters).  OPTIONAL groups may be absent (no such launch in the profiled command).
GROUPS = {
    "harp_shade_bwd": (("shade_bwd_wave_kernel", "shade_bwd_face_kernel", "shade_kernel<true>"),),
    "harp_shade_fwd": (("shade_kernel<false>",),),
    # face_setup / bin_faces / order_tiles run once per view: their per-launch averages are over both views already
    "raster_cam_fwd(setup+bin+raster)": (("raster_kernel<1,",), ("face_setup_kernel",), ("bin_faces_kernel", "expand_bits_kernel"), ("order_tiles_kernel",)),
    "raster_light_fwd(setup+bin+raster)": (("raster_kernel<0,",), ("face_setup_kernel",), ("bin_faces_kernel", "expand_bits_kernel"), ("order_tiles_kernel",)),
    "raster_cam_fwd(raster kernel only)": (("raster_kernel<1,",),),
    "raster_light_fwd(raster kernel only)": (("raster_kernel<0,",),),
    "harp_silhouette_bwd": (("raster_kernel<2,",),),
    "harp_depth_bwd": (("depth_bwd_kernel",),),
    "harp_texel_reduce": (("texel_reduce_kernel",), ("texel_finish_kernel",), ("texel_counters_clear_kernel",)),
}
OPTIONAL = {"harp_shade_fwd", "harp_texel_reduce"}       # not launched in the fitting loop's fused-loss mode / in the table form of the shader backward


def main(fetch_db, write_db, cmd):
    f, w = per_launch(fetch_db, "FETCH_SIZE"), per_launch(write_db, "WRITE_SIZE")
    out = {"_source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) -- {cmd}; per-launch averages over the LAST {LAST} "
                      "launches of every kernel (= the profiled steps; the launches that render bench.py's targets are left out); "
                      "HBM bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950: FETCH_SIZE under-reports coalesced reads by 2x, calibrated on "
                      "adam_dev_kernel; WRITE_SIZE calibrated on the raster outputs).  tools/make_traffic_json.py"}
    try:
        import os
        import subprocess
        out["_head"] = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=10,
                                      cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip() or "unknown (no git on this box)"
    except Exception:
        out["_head"] = "unknown"
    detail = {}
    names = set(f) | set(w)
    for key, parts in GROUPS.items():
        b = 0.0
        for alts in parts:
            hits = [n for n in names if any(a in n for a in alts)]
            if not hits:
                if key in OPTIONAL:
                    continue
                sys.exit(f"make_traffic_json: no profiled kernel matches {alts} (group {key!r}); kernels seen: {sorted(n[:50] for n in names)}")
            for name in hits:
                fb, wb = f.get(name, (0, 0))[0], w.get(name, (0, 0))[0]
                b += 2 * fb * 1024 + wb * 1024
                detail[name[:60]] = {"fetch_KiB": round(fb, 1), "write_KiB": round(wb, 1), "launches": f.get(name, (0, 0))[1]}
        if b == 0.0 and key not in OPTIONAL:
            sys.exit(f"make_traffic_json: group {key!r} has zero bytes")
        out[key] = int(b)
    out["_per_kernel"] = detail
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
