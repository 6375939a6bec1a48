"""bench.py — frames/s of HARP's render + loss + backward + Adam inner step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N>1: launched by torch.distributed.run, one rank per GPU)

Workload (BASELINE.json configs[2]/[3], SURVEY.md §8d "C3/C4"): 256-frame synthetic sequence, 512x512, subdivided MANO
hand mesh (3093 v / 6152 f / 3327 uv), self-shadow on, per-vertex displacement + texture/normal-map optimisation stage
(coarse AND appearance terms active: silhouette, keypoint, displacement reg, laplacian, normal consistency, ARAP,
photometric, albedo reg, normal-map reg; VGG excluded per SURVEY.md §8f), 32 frames per GPU per step (weak scaling),
one flat all-reduce of the parameter gradients over RCCL before the replicated dense Adam step.
A "step" = LBS -> mesh prep -> 3 rasterisations -> shade -> losses -> backward -> [all-reduce] -> 2x Adam on 32 frames/GPU.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # this pool's driver only supports dmabuf IPC (RCCL across processes)

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from harp_amd import synth  # noqa: E402
from harp_amd.engine import FitEngine  # noqa: E402

T_FRAMES, S, B_PER_GPU = 256, 512, 32
HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def erode(mask, iters=2):
    """3x3 erosion x2 (utils/data_util.py:17-20) on the device."""
    m = mask[:, None]
    for _ in range(iters):
        m = -torch.nn.functional.max_pool2d(-m, 3, stride=1, padding=1)
    return m[:, 0]


def build_engine(rank, world, device, T=T_FRAMES, img=S, B=B_PER_GPU, seed=0, kind="hand"):
    """kind "hand": subdivided MANO hand (C2/C3/C4); "arm": SMPL-X right-arm mesh 4083 v / 8128 f with the kinematic-tree LBS (C5)"""
    if kind == "hand":
        tpl = synth.load_template("hand")
        topo = synth.build_topology(tpl["faces0"], 778)
        model = synth.make_mano_model(tpl, seed=seed)
        seq, focal = synth.make_sequence(model, T, img, seed=seed)
        kw = {}
    else:
        tpl = synth.load_template("arm")
        topo = synth.build_topology(tpl["faces0"], 1026)
        model = synth.make_smplx_arm_model(tpl, seed=seed)
        focal = 1000.0 * img / 224.0
        ga = torch.Generator().manual_seed(seed + 1)
        c = model["v_template"].mean(0)
        seq = dict(pose=torch.randn(T, 45, generator=ga) * 0.15, rot=torch.randn(T, 3, generator=ga) * 0.2, trans=torch.zeros(T, 3),
                   shape=torch.randn(T, 10, generator=ga) * 0.3,
                   cam=torch.tensor([[2 * focal / (img * 1.6), -float(c[0]), -float(c[1])]]).repeat(T, 1) + torch.randn(T, 3, generator=ga) * 0.005)
        kw = dict(use_arm=True, opt_arm_pose=True)
    seq["joints"] = torch.zeros(T, 21, 3)
    eng = FitEngine(model, topo, tpl["verts_uvs"], tpl["faces_uvs"], tpl["uv_mask"].astype(np.float32) / 255.0, seq, img, focal, B,
                    device=device, rank=rank, world_size=world, seed=seed, **kw)
    # ---- synthetic targets: render a perturbed "ground-truth" parameter set with the engine itself
    Tl = T // world
    lo = rank * Tl
    g = torch.Generator().manual_seed(seed + 7)
    saved = {k: eng.params[k].clone() for k in ("pose", "cam", "texture", "normal_map", "verts_disps", "shape")}
    with torch.no_grad():
        eng.params["pose"].add_((torch.randn(T, 45, generator=g) * 0.05).to(device))
        eng.params["cam"][:, 1:].add_((torch.randn(T, 2, generator=g) * 0.004).to(device))
        eng.params["shape"].add_((torch.randn(10, generator=g) * 0.3).to(device))
        eng.params["verts_disps"].copy_((torch.randn(eng.topo.V, 1, generator=g) * 0.0008).to(device))
        tex = torch.nn.functional.interpolate(torch.rand(1, 3, 32, 32, generator=g), size=512, mode="bilinear")[0].permute(1, 2, 0)
        eng.params["texture"].copy_((0.35 + 0.5 * tex)[None].to(device))
    y_true = torch.empty(Tl, img, img, 3, device=device)
    y_sil = torch.empty(Tl, img, img, device=device)
    joints = torch.empty(T, eng.n_joints, 3, device=device)
    eng.set_stage(False, True)
    eng.y_true, eng.y_sil, eng.y_sil_col = y_true, y_sil, y_sil      # placeholders (losses are ignored here)
    for s0 in range(0, T, B):
        fid = torch.arange(s0, s0 + B, dtype=torch.int32)
        eng.fid.copy_(fid.to(device))
        eng.tfid.zero_()
        eng.forward_backward(coarse=True, app=True)
        joints[s0:s0 + B] = eng.s["joints_mm"]
        if lo <= s0 < lo + Tl:
            y_true[s0 - lo:s0 - lo + B] = eng.s["rgb"]
            y_sil[s0 - lo:s0 - lo + B] = (eng.s["alpha"] > 0.5).float()
    torch.cuda.synchronize()
    with torch.no_grad():
        for k, v in saved.items():
            eng.params[k].copy_(v)
    eng.init_joints = (joints[:, :21] + torch.randn(T, 21, 3, generator=g).to(device) * 2.0).contiguous()    # METRO-like noisy anchors (mm)
    eng.set_targets(y_true, y_sil, erode(y_sil), frame_offset=lo)
    eng.compute_reference_mesh()
    eng.g_buf.zero_()
    return eng, focal


def algorithmic_bytes(eng):
    """A_frame and A_step for this workload (fp32, int32 indices, shadow on)."""
    V, F, VT = eng.topo.V, eng.topo.F, eng.topo.verts_uvs.shape[0]
    S2, T2 = eng.S * eng.S, eng.Ht * eng.Wt
    geom = V * 12 + F * 12 + VT * 8 + F * 12
    targets, outs, depth = S2 * 20, S2 * 16, S2 * 4
    a_frame = 2 * geom + 2 * targets + 2 * outs + 3 * depth + V * 12
    n_param = 2 * T2 * 3 + V + 14
    a_step = 4 * T2 * 12 + 28 * n_param
    return a_frame, a_step, dict(geom=geom, S2=S2, V=V, F=F)


def kernel_roofline(eng, steps, overlap=False):
    """Duration of the dominant kernel, measured with HIP events on the launch stream in an eager (non-graph) re-run of the
    same steps right after the timed region; algorithmic bytes per launch from SURVEY.md §8(d).
    overlap=False: one stream, every event pair brackets exactly one kernel group running alone.  overlap=True: the step's real
    two-stream schedule — the pair then measures the group IN SITU, next to whatever the second stream runs at that moment."""
    from harp_amd import _lib
    L = _lib.lib()
    # (the fitting loop calls the light-view pass and its backward through the variants that keep their images across steps)
    alias = {"harp_rasterize_fwd_keep": "raster_light", "harp_depth_bwd_consume": "harp_depth_bwd", "harp_depth_nmap_bwd": "harp_depth_bwd", "harp_depth_bwd_tiles": "harp_depth_bwd", "harp_depth_bwd_riders": "harp_depth_bwd"}
    alias.update({"harp_texel_finish": "harp_texel_reduce"})        # (reduce + finish: the second half of the shader backward's texel gradients)
    names = ["harp_rasterize_fwd", "harp_rasterize_l1_fwd", "harp_shade_fwd", "harp_shade_bwd", "harp_silhouette_bwd", "harp_depth_bwd", "harp_texel_reduce"] + list(alias)
    rec = {n: [] for n in names}
    orig = {}

    class Timed:
        def __init__(self, name, fn):
            self.name, self.fn = name, fn

        def __call__(self, *a):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = self.fn(*a)
            e1.record()
            # harp_rasterize_fwd: argument 6 is `soft` (bit 0 = camera view with the fused soft silhouette, else the light-view depth pass)
            key = alias.get(self.name) or (self.name if "rasterize" not in self.name else ("raster_cam" if (a[6] & 1) else "raster_light"))
            rec.setdefault(key, []).append((e0, e1))
            return r

    for n in names:
        orig[n] = getattr(L, n)
        setattr(L, n, Timed(n, orig[n]))
    try:
        eng.overlap = overlap        # False = one stream: event pairs then bracket exactly one kernel group each
        for i in range(steps):
            fid = (torch.arange(eng.B) + i * eng.B) % (eng.T // eng.world) + eng.target_offset
            eng.step(fid, True, True, use_graph=False)
        torch.cuda.synchronize()
    finally:
        eng.overlap = True
        for n in names:
            setattr(L, n, orig[n])
    ms = {n: [a.elapsed_time(b) for a, b in v] for n, v in rec.items()}
    out = {"raster_cam_fwd(setup+bin+raster)": float(np.mean(ms["raster_cam"])), "raster_light_fwd(setup+bin+raster)": float(np.mean(ms["raster_light"]))}
    for n in names[2:6]:
        if ms.get(n):
            out[n] = float(np.mean(ms[n]))
    if ms.get("harp_texel_reduce"):                # two calls per step (reduce, finish): their sum
        out["harp_texel_reduce"] = float(np.sum(ms["harp_texel_reduce"])) / steps
    return out


# kernel-name substrings of the event-timed groups (the in-graph durations of a group = the sum over its kernels; the three set-up kernels
# run once per view, their per-launch averages are over both views)
_GROUP_KERNELS = {"raster_cam_fwd(setup+bin+raster)": ("raster_kernel<1,", "face_setup_kernel", "bin_faces_kernel", "expand_bits_kernel", "order_tiles_kernel"),
                  "raster_light_fwd(setup+bin+raster)": ("raster_kernel<0,", "face_setup_kernel", "bin_faces_kernel", "expand_bits_kernel", "order_tiles_kernel"),
                  "harp_shade_fwd": ("shade_kernel<false>",), "harp_shade_bwd": ("shade_bwd_wave_kernel",),
                  "harp_silhouette_bwd": ("raster_kernel<2,",), "harp_depth_bwd": ("depth_bwd_kernel",),
                  "harp_texel_reduce": ("texel_reduce_kernel", "texel_finish_kernel", "texel_counters_clear_kernel")}


def _child(cmd_tail, prof_args, out_dir, timeout=240):
    """one `rocprofv3 <prof_args> -- python bench.py <cmd_tail>` child (outside the timed region); returns the rocpd database path or None"""
    import glob
    import shutil
    import subprocess
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None
    env = dict(os.environ, HARP_BENCH_CHILD="1", TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [exe, *prof_args, "-d", out_dir, "-o", "run", "--", sys.executable, os.path.join(ROOT, "bench.py"), *cmd_tail]
    try:
        r = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=timeout)
    except Exception as e:                                       # noqa: BLE001
        print(f"[bench] profiling child failed: {type(e).__name__}: {e}", file=sys.stderr)
        return None
    dbs = glob.glob(os.path.join(out_dir, "**", "*.db"), recursive=True)
    if r.returncode != 0 or not dbs:
        print(f"[bench] profiling child rc={r.returncode}: {r.stderr[-400:]}", file=sys.stderr)
        return None
    return dbs[0]


def profiled_in_graph(steps=40, warmup=8):
    """In-graph kernel durations of THIS command, the way profiles/ records them: a child `rocprofv3 --kernel-trace -- python bench.py
    --steps 40 --warmup 8 --no-cpu-baseline --no-extras --no-roofline` (graph replays, three streams), average over the last `steps`
    launches of every kernel (= the timed replays; the launches that render the targets come first) -> {kernel name: (avg_us, launches)},
    or None when rocprofv3 is not available / the child fails."""
    import shutil
    import sqlite3
    import tempfile
    out_dir = tempfile.mkdtemp(prefix="harp_prof_", dir="/tmp")
    try:
        db = _child(["--steps", str(steps), "--warmup", str(warmup), "--no-cpu-baseline", "--no-extras", "--no-roofline"], ["--kernel-trace"], out_dir)
        if db is None:
            return None
        cur = sqlite3.connect(db).cursor()
        cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
        name = "name" if "name" in cols else [c for c in cols if "name" in c][0]
        per = {}
        for n, t0, t1 in cur.execute(f"select {name}, start, end from kernels order by start"):
            per.setdefault(n, []).append((t1 - t0) / 1e3)
        out = {}
        for n, d in per.items():
            k = steps * (2 if any(t in n for t in ("face_setup_kernel", "bin_faces_kernel", "expand_bits_kernel", "order_tiles_kernel")) else 1)
            keep = d[-k:]
            out[n] = (float(np.mean(keep)), len(keep))
        return out
    except Exception as e:                                       # noqa: BLE001
        print(f"[bench] kernel-trace pass unusable: {type(e).__name__}: {e}", file=sys.stderr)
        return None
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)


def profiled_traffic():
    """HBM bytes per launch of the kernel groups from two separate PMC passes over this command (FETCH_SIZE, WRITE_SIZE; counters only
    with --kernel-trace, 3 eager steps each; units and the gfx950 2x FETCH correction as in tools/make_traffic_json.py, which this
    reuses) -> dict like profiles/traffic_latest.json, or None."""
    import contextlib
    import io
    import shutil
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import make_traffic_json as mt
    except Exception:                                            # noqa: BLE001
        return None
    tail = ["--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--no-roofline", "--no-graph"]
    dirs, dbs = [], []
    try:
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            d = tempfile.mkdtemp(prefix=f"harp_pmc_{c}_", dir="/tmp")
            dirs.append(d)
            db = _child(tail, ["--kernel-trace", "--pmc", c], d)
            if db is None:
                return None
            dbs.append(db)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            mt.main(dbs[0], dbs[1], "python bench.py " + " ".join(tail))
        return json.loads(buf.getvalue())
    except (Exception, SystemExit) as e:                         # noqa: BLE001
        print(f"[bench] PMC passes unusable: {type(e).__name__}: {e}", file=sys.stderr)
        return None
    finally:
        for d in dirs:
            shutil.rmtree(d, ignore_errors=True)


# measured VALU issue rate of a wave64 instruction with >= 2 waves per SIMD (profiles/r04_valu_issue_micro.txt): 2.5 cycles
VALU_ISSUE_CYCLES = 2.5
SIMDS, CLOCK_HZ = 1024, 2.4e9
ATOMIC_RATE = 330e9        # memory-side float atomics per second, consecutive addresses (tools/dev/micro/glb_atomics)


def profiled_issue():
    """VALU wave-instructions per launch of every kernel from one PMC pass over this command (SQ_INSTS_VALU; counters only with
    --kernel-trace, 3 eager steps): the counter's rows of one dispatch (one per shader-engine slice) are summed, the last launches of every
    kernel averaged -> {kernel name: wave-instructions per launch}, or None."""
    import shutil
    import sqlite3
    import tempfile
    tail = ["--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--no-roofline", "--no-graph"]
    d = tempfile.mkdtemp(prefix="harp_pmc_valu_", dir="/tmp")
    try:
        db = _child(tail, ["--kernel-trace", "--pmc", "SQ_INSTS_VALU"], d)
        if db is None:
            return None
        cur = sqlite3.connect(db).cursor()
        cols = [r[1] for r in cur.execute("pragma table_info(pmc_events)")]
        name = [c for c in cols if c in ("name", "kernel_name")][0]
        val = "value" if "value" in cols else "counter_value"
        per = {}
        for n, disp, v in cur.execute(f"select {name}, dispatch_id, sum({val}) from pmc_events group by {name}, dispatch_id order by dispatch_id"):
            per.setdefault(n, []).append(float(v))
        return {n: float(np.mean(v[-3:])) for n, v in per.items()}
    except Exception as e:                                       # noqa: BLE001
        print(f"[bench] SQ_INSTS_VALU pass unusable: {type(e).__name__}: {e}", file=sys.stderr)
        return None
    finally:
        shutil.rmtree(d, ignore_errors=True)


def _graph_rate(e, steps, warmup, coarse=True, app=True):
    """frames/s of graph-replayed scheduled steps of engine `e` (single GPU, outside the timed region of the headline)"""
    B, T = e.B, e.T
    e.set_schedule(torch.stack([(torch.arange(B) + i * B) % T for i in range(warmup + steps)]).to(torch.int32))
    for _ in range(warmup):
        e.step(None, coarse, app)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        e.step(None, coarse, app)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"frames_per_s": B * steps / dt, "ms_per_step": dt / steps * 1e3, "frames_per_step": B, "steps": steps}


def extra_rates(eng, device, steps=100, warmup=30, vgg_weights="random"):
    """Rates of the other BASELINE.json configurations and modes (not the headline `value`): the same step with the rendered image
    materialised like the reference's y_pred (keep_image=True), C2 at the reference's batch size 18, and C5's per-GPU share (SMPL-X
    arm mesh at 1024x1024, 32 frames / GPU).  Graph-replayed steps, barrier-free single GPU, synthetic targets rendered by the engine.
    (30 warm-up steps: the first ~50 ms after an engine is built run 5 % slower — the device idles while the host renders the targets
    — and a 6-step warm-up put that ramp into the 40 timed steps: 1.886 ms / step against 1.79 for every later measurement of the same
    engine, tools/dev/gpu_c5_rate.py in round 4.)"""
    rate = lambda e: _graph_rate(e, steps, warmup)
    out = {}
    # the other two stages of the reference's schedule (configs["training_stage"], optimize_sequence.py:426-441) on the headline's engine and
    # scene: geometry only (silhouette, key points, mesh regularisers -> opt_coarse) and appearance only (photometric + texture regularisers
    # -> opt_app) — the latter also as the fitting API runs it (`lean_app_stage`: without the geometry gradients nothing reads)
    keep = (eng.keep_image, eng.lean_app_stage)
    eng.keep_image = False
    out["C3_stage_geometry_only"] = _graph_rate(eng, steps, warmup, True, False)
    out["C3_stage_appearance_only"] = _graph_rate(eng, steps, warmup, False, True)
    eng.lean_app_stage = True
    out["C3_stage_appearance_only_lean"] = _graph_rate(eng, steps, warmup, False, True)
    eng.keep_image, eng.lean_app_stage = keep
    e = build_engine(0, 1, device, T=72, img=S, B=18)[0]
    e.keep_image = False
    out["C2_reference_batch_18"] = rate(e)
    del e
    torch.cuda.empty_cache()
    e = build_engine(0, 1, device, T=32, img=1024, B=32, kind="arm")[0]
    e.keep_image = False
    # (this engine's targets take the host ~2 s to render; its first ~150 ms of steps run up to 5 % slower than every later measurement —
    #  1.886, 1.817, then 1.788 ms / step for three consecutive 40-step windows in tools/dev/gpu_c5_rate.py — hence the longer warm-up)
    out["C5_arm_1024_per_gpu_share"] = dict(_graph_rate(e, steps, max(warmup, 100)), mesh="SMPL-X right arm 4083v/8128f, kinematic-tree LBS")
    del e
    torch.cuda.empty_cache()
    if vgg_weights is not None:
        out["C3_with_perceptual_term"] = perceptual_rate(device, vgg_weights)
    out["C3_one_rank_allreduce_in_graph"] = one_rank_allreduce_cost(eng, steps, warmup)
    return out


def one_rank_allreduce_cost(eng, steps, warmup):
    """What the N > 1 step adds to the single-GPU step BEFORE any wire time: the headline engine on a 1-rank RCCL communicator — the two
    captured harp_allreduce_flat nodes (map gradients early on the communication stream, the remainder in front of Adam), their fork / join
    edges, and the engine in its N > 1 configuration (table form of the shader backward: the map gradients are final ~60 us earlier).
    DESIGN.md 5 builds its 2 / 4 / 8-GPU expectation on this figure."""
    try:
        from harp_amd.dist import RcclComm
        base = _graph_rate(eng, steps, warmup)
        keep = (eng.texel_records, eng.force_allreduce)
        eng.texel_records = False
        table = _graph_rate(eng, steps, warmup)
        comm = RcclComm(0, 1, RcclComm.unique_id())
        eng.force_allreduce = True
        eng.set_comm(comm)
        with_comm = _graph_rate(eng, steps, warmup)
        torch.cuda.synchronize()
        eng.set_comm(None)
        comm.destroy()
        eng.texel_records, eng.force_allreduce = keep
        eng._graphs = {}
        return {"ms_per_step_single_gpu_default": base["ms_per_step"], "ms_per_step_table_form_no_collective": table["ms_per_step"],
                "ms_per_step_with_two_captured_1rank_allreduce_nodes": with_comm["ms_per_step"],
                "added_us": (with_comm["ms_per_step"] - table["ms_per_step"]) * 1e3, "bucket_bytes": int(eng.opt_span[1]) * 4, "steps": steps}
    except Exception as e:                                       # noqa: BLE001  (no RCCL on this box: the figure is reported as missing, not guessed)
        return {"error": f"{type(e).__name__}: {e}"}


# VGG16 features[0:23] at 512x512: 3x3 convolutions (Cin, Cout, H) -> 2 * 9 * Cin * Cout * H * H flop each
_VGG_CONVS = ((3, 64, 512), (64, 64, 512), (64, 128, 256), (128, 128, 256), (128, 256, 128), (256, 256, 128), (256, 256, 128), (256, 512, 64),
              (512, 512, 64), (512, 512, 64))


def perceptual_rate(device, weights, steps=6, warmup=2):
    """The same step with the reference's default VGG feature term on (optimize_sequence.py:404-405, 419, 546-547):
    the ten 3x3 convolutions, their data gradients and the ReLU / max-pool / L1 glue as the HIP kernels of csrc/conv.hip, captured into
    the step's hipGraph with every other launch.  Target features are cached in HBM (the targets do not change during a fit), so a step is
    one VGG forward + one backward-data pass over the B rendered images.  Both arithmetic modes are timed: float32 MFMA
    (v_mfma_f32_32x32x2_f32, priced against its 157.3 TFLOP/s peak) and the three-term bf16 split (3 v_mfma_f32_32x32x16_bf16 per product
    block, priced against the 2.5 PFLOP/s dense bf16 peak at 3x the flop count).  weights: "random" (seeded filters: the timing does not
    depend on the values) or the path of a torchvision vgg16 state dict."""
    from harp_amd.model.vgg import Vgg16Features
    fwd = sum(2.0 * 9 * ci * co * h * h for ci, co, h in _VGG_CONVS) * B_PER_GPU
    flop = 2.0 * fwd                                   # forward + backward-data (the filters are frozen: no weight gradients)
    vgg = Vgg16Features(layers_weights=[1, 1 / 16, 1 / 8, 1 / 4, 1], weights=weights)
    res = {"frames_per_step": B_PER_GPU, "steps": steps, "conv_tflop_per_step_full_images": flop / 1e12,
           "filters": "random (seeded)" if weights == "random" else os.path.basename(str(weights))}
    for name, prec, peak, mult, bounded in (("f32_mfma", 0, 157.3e12, 1.0, True), ("bf16x3_split", 1, 2.5e15, 3.0, True),
                                            ("f32_mfma_full_images", 0, 157.3e12, 1.0, False)):
        e = build_engine(0, 1, device, T=B_PER_GPU, img=S, B=B_PER_GPU)[0]
        e.keep_image = False
        t0 = time.perf_counter()
        e.set_perceptual(vgg, precision=prec, bounded=bounded)
        torch.cuda.synchronize()
        t_set = time.perf_counter() - t0
        e.set_schedule(torch.arange(B_PER_GPU).reshape(1, -1).to(torch.int32))
        for _ in range(warmup):
            e.step(None, True, True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            e.step(None, True, True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        # bounded mode: the stack runs in the tiles (16 px a side, 8 at S/4 and S/8) the mask's support reaches (harp_amd/model/vgg_hip.active_tiles): executed flop =
        # per-level flop x the level's share of active tiles over the batch's frames
        done, tiles = flop, None
        if e._vgg_bound is not None:
            tiles = [float(b[2].float().sum()) * b[6] ** 2 / (b[0].shape[0] * float(S >> lv) ** 2) for lv, b in enumerate(e._vgg_bound)]      # share of the image's area
            done = 2.0 * B_PER_GPU * sum(2.0 * 9 * ci * co * h * h * tiles[{512: 0, 256: 1, 128: 2, 64: 3}[h * 512 // S]] for ci, co, h in _VGG_CONVS)
        res[name] = {"frames_per_s": B_PER_GPU / dt, "ms_per_step": dt * 1e3, "in_hipgraph": bool(e._graphs),
                     "active_tile_share_per_level": tiles, "conv_tflop_executed": done / 1e12,
                     "roofline": {"bound": "mfma", "achieved": mult * done / dt / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s",
                                  "frac": mult * done / dt / peak},
                     "set_up_s(filter packing + target features)": t_set, "vgg_loss": e.losses().get("vgg")}
        del e
        torch.cuda.empty_cache()
    res["frames_per_s"] = res["f32_mfma"]["frames_per_s"]
    res["ms_per_step"] = res["f32_mfma"]["ms_per_step"]
    return res


def cpu_baseline(seed=0, frames_per_step=4):
    """The CPU oracle (a restatement of the reference path: materialised (B,S,S,K) fragments, torch autograd, torch.optim.Adam) timed
    on this box's host cores on the bounded sample SURVEY.md §8(d) names: the C2/C3 step reduced to 4 frames per step at 512x512 (so
    the dense Adam update over 1.58 M parameters and the frame-independent regularisers are shared by 4 frames, as they are shared by
    18 / 32 in the real configurations), and config C1 in full (one 256x256 frame, raw 778-vertex MANO mesh, silhouette term only)."""
    from oracle import harp_ref as H
    # many-core hosts oversubscribe badly on these small ops (256 threads: 306 s/step vs 3.9 s with 8) -> cap at 16
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    tpl = synth.load_template("hand")
    topo_np = synth.build_topology(tpl["faces0"], 778)
    model_np = synth.make_mano_model(tpl, seed=seed)
    model = {k: torch.from_numpy(v) for k, v in model_np.items()}
    topo = {k: torch.from_numpy(np.asarray(v)).long() if isinstance(v, np.ndarray) else v for k, v in topo_np.items()}
    n = int(frames_per_step)
    T = 2 * n
    seq, focal = synth.make_sequence(model_np, T, S, seed=seed)
    P = dict(pose=seq["pose"], cam=seq["cam"], verts_disps=torch.zeros(3093, 1), shape=seq["shape"].mean(0),
             light_positions=torch.tensor(((-0.5, -0.5, -0.5),)).repeat(T, 1), amb_ratio=torch.tensor(0.4),
             texture=torch.tensor([232, 190, 172]).repeat(1, 512, 512, 1) / 255., normal_map=torch.tensor([0., 0., 1.]).repeat(1, 512, 512, 1),
             rot=seq["rot"], trans=seq["trans"])
    P = {k: v.clone().requires_grad_() for k, v in P.items()}
    P.update(verts_uvs=torch.from_numpy(tpl["verts_uvs"]), faces_uvs=torch.from_numpy(tpl["faces_uvs"]).long(),
             uv_mask=torch.from_numpy(tpl["uv_mask"]).double() / 255, init_joints=torch.zeros(T, 21, 3))
    tg = dict(y_true=torch.rand(T, S, S, 3), y_sil=(torch.rand(T, S, S) > 0.5).float(), y_sil_col=(torch.rand(T, S, S) > 0.5).float())
    opt_c = torch.optim.Adam([{"params": [P["pose"], P["cam"]], "lr": 1e-3}, {"params": [P["verts_disps"], P["shape"]], "lr": 1e-3}])
    opt_a = torch.optim.Adam([P["light_positions"], P["amb_ratio"], P["texture"], P["normal_map"]], lr=1e-2)
    with torch.no_grad():
        _, rv = H.prepare_mesh(P, torch.tensor([0]), model, topo)

    def timed_steps(k):
        times = []
        for it in range(k):
            t0 = time.time()
            fid = (torch.arange(n) + it * n) % T
            da = torch.normal(0, 1.0, (512, 512, 2)).to(torch.int).long()
            dn = torch.normal(0, 2.0, (512, 512, 2)).to(torch.int).long()
            _, total, _ = H.step_losses(P, fid, model, topo, tg, S, focal, rv, da, dn)
            opt_c.zero_grad(); opt_a.zero_grad()
            total.backward()
            opt_c.step(); opt_a.step()
            times.append(time.time() - t0)
        return float(np.median(times[1:]))

    sec = timed_steps(5)
    torch.set_num_threads(1)                             # also a single-core figure
    sec1 = timed_steps(2)
    torch.set_num_threads(cores)
    # ---- C1 in full: one 256x256 frame, the un-subdivided MANO mesh, silhouette L1 only, Adam over the coarse group
    raw_np = synth.build_raw_topology(tpl["faces0"], 778)
    raw = {k: torch.from_numpy(np.asarray(v)).long() if isinstance(v, np.ndarray) else v for k, v in raw_np.items()}
    seq1, focal1 = synth.make_sequence(model_np, 1, 256, seed=seed)
    Q = {k: seq1[k].clone().requires_grad_() for k in ("pose", "cam", "rot", "trans")}
    Q.update(shape=seq1["shape"].mean(0).clone().requires_grad_(), verts_disps=torch.zeros(778, 1, requires_grad=True))
    y1 = (torch.rand(1, 256, 256) > 0.5).float()
    opt1 = torch.optim.Adam([{"params": [Q["pose"], Q["cam"]], "lr": 1e-3}, {"params": [Q["verts_disps"], Q["shape"]], "lr": 1e-3}])
    t_c1 = []
    for it in range(6):
        t0 = time.time()
        _, v1 = H.prepare_mesh(Q, torch.tensor([0]), model, raw)
        loss = 7.0 * torch.nn.functional.l1_loss(y1, H.render_silhouette(v1, raw["faces"], Q["cam"][:1], 256, focal1))
        opt1.zero_grad(); loss.backward(); opt1.step()
        t_c1.append(time.time() - t0)
    c1 = float(np.median(t_c1[1:]))
    return {"value": n / sec, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"oracle/harp_ref.step_losses + autograd + torch.optim.Adam on the C2/C3 step reduced to {n} frames/step at {S}x{S} (SURVEY.md §8d; K=50 "
                      f"silhouette fragments materialised, all terms but VGG, dense Adam shared by the {n} frames), median of 4 steps after 1 warm-up "
                      f"(~{5 * sec + 2 * sec1 + 6 * c1:.0f} s of CPU work in total), torch.set_num_threads({cores})",
            "s_per_step": sec,
            "single_core": {"value": n / sec1, "unit": "frames/s", "cores": 1, "sample": "same step, 1 timed after 1 warm-up"},
            "C1": {"value": 1.0 / c1, "unit": "frames/s", "cores": cores,
                   "sample": "config C1 in full: one 256x256 frame, raw MANO mesh 778v/1538f, silhouette L1 only, Adam; median of 5 steps after 1 warm-up"}}


class _StdoutToStderr:
    """RCCL prints a version banner on fd 1 when the communicator is created; keep stdout for the ONE JSON line"""
    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *a):
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        os.dup2(self.saved, 1)
        os.close(self.saved)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the rates of the other configurations (keep_image, B=18, C5 arm 1024)")
    ap.add_argument("--no-roofline", action="store_true", help="skip the per-kernel event timing and the keep_image rate (profiling runs: only the timed steps)")
    ap.add_argument("--no-profile", action="store_true", help="do not start rocprofv3 children for the in-graph kernel durations / HBM traffic of the "
                    "roofline entry (they run outside the timed region; without them the entry falls back to HIP-event timing and the committed traffic file)")
    ap.add_argument("--vgg-weights", default="random", help="filters of the perceptual-term entry of `extras`: 'random' (default, timing only), "
                    "'none' (skip the entry) or the path of torchvision's vgg16 state dict (vgg16-397923af.pth)")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N")
    # development aid: HARP_ALL_ON_GPU0=1 + HARP_DIST_BACKEND=gloo runs an N-process job on a ONE-GPU box (every rank on cuda:0, the
    # collectives through gloo) to exercise the N > 1 control flow end to end; its timing means nothing and is marked as such
    shared_gpu = os.environ.get("HARP_ALL_ON_GPU0") == "1"
    backend = os.environ.get("HARP_DIST_BACKEND", "nccl")
    if shared_gpu:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    force_dist = os.environ.get("HARP_FORCE_DIST") == "1"      # exercise the RCCL code path (init, barrier, all-reduce, eager steps) on 1 GPU
    if world > 1 or force_dist:
        import torch.distributed as dist
        if force_dist and "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        with _StdoutToStderr():
            import datetime
            dist.init_process_group(backend, timeout=datetime.timedelta(seconds=int(os.environ.get("HARP_DIST_TIMEOUT_S", "300"))),
                                    **({"device_id": device} if backend == "nccl" else {}))
            dist.barrier()                               # communicator creation happens at the first collective
            torch.cuda.synchronize()
    eng, focal = build_engine(rank, world, device)
    eng.force_allreduce = force_dist
    eng.keep_image = False                      # the step consumes the rendered image inside the shader (fused L1): it is not written out
    eng.graph_collectives = os.environ.get("HARP_GRAPH_COLLECTIVES", "0") == "1"
    comm = None
    if (world > 1 or force_dist) and backend == "nccl" and not shared_gpu and os.environ.get("HARP_NO_RCCL_COMM") != "1":
        # production N > 1 path: RCCL called directly through the C ABI on the step's own streams (harp_allreduce_flat), so the
        # collective is a node of the same hipGraph as the kernels; torch.distributed only carries the 128-byte communicator id
        from harp_amd.dist import negotiate_comm
        # never exercised on more than one rank in the build environment (1-GPU boxes only).  harp_amd.dist.negotiate_comm: a LOCAL pre-flight
        # (dlopen + dlsym of RCCL, ncclGetUniqueId — no peer involved) agreed over the process group, then the communicator, agreed again: if
        # ANY rank cannot have it, every rank falls back to torch.distributed's all-reduce with eager steps and the line says so
        # ("collective").  BEST EFFORT: a rank that dies INSIDE ncclCommInitRank or inside a captured all-reduce still hangs its peers until
        # the process group's timeout — RCCL's own failure mode.  HARP_RCCL_DEBUG=1 forces the eager fallback.
        with _StdoutToStderr():
            comm = negotiate_comm(device)
        if comm is not None:
            eng.set_comm(comm)
    Tl = eng.T // world

    # the whole frame schedule lives on the device (the reference's DataLoader hands over host tensors every step)
    sched = torch.stack([(torch.arange(eng.B) + i * eng.B) % Tl + eng.target_offset for i in range(args.warmup + args.steps)]).to(torch.int32).to(device)
    eng.set_schedule(sched)                     # step(None, ...) advances through it inside the captured graph: zero host-side copies

    def sync():
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    use_graph = not args.no_graph
    graph_fallback = None
    try:
        for i in range(args.warmup):
            eng.step(None, True, True, use_graph=use_graph)
        torch.cuda.synchronize()
        ok = 1
    except Exception as e:                                    # noqa: BLE001
        graph_fallback = f"{type(e).__name__}: {e}"
        ok = 0
    if world > 1 or force_dist:
        # all ranks take the same path: a capture of the step (with its collective) that fails anywhere switches every rank to eager steps
        flag = torch.tensor([ok], device=device, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = int(flag.item())
    if not ok:
        if not use_graph:
            raise RuntimeError(f"warm-up steps failed: {graph_fallback}")
        print(f"[bench] rank {rank}: graph-replayed steps failed ({graph_fallback}); eager steps", file=sys.stderr)
        # (`torch.cuda.graph`'s context manager ends a capture that failed half-way on its way out; the partial graph is dropped with the
        # engine's graph table below) — drain the device before the eager steps use the same streams
        torch.cuda.synchronize()
        use_graph, eng._graphs = False, {}
        graph_fallback = graph_fallback or "another rank failed"
        for i in range(args.warmup):
            eng.step(None, True, True, use_graph=False)
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        eng.step(None, True, True, use_graph=use_graph)
    sync()
    dt = time.perf_counter() - t0
    dt_local = dt
    if world > 1 or force_dist:
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    per_rank_ms, allreduce = None, None
    if world > 1 or force_dist:
        tmax = torch.tensor([dt_local], device=device, dtype=torch.float64)
        allt = [torch.zeros_like(tmax) for _ in range(world)]
        dist.all_gather(allt, tmax)
        per_rank_ms = [float(t.item()) / args.steps * 1e3 for t in allt]
        # the collective on its own (outside the timed region): the flat gradient bucket, K back-to-back all-reduces on one stream
        bucket = eng.g_buf[eng.opt_span[0]:eng.opt_span[0] + eng.opt_span[1]]
        K = 20
        def one():
            if comm is not None:
                comm.allreduce(bucket)
            else:
                dist.all_reduce(bucket)
        for _ in range(3):
            one()
        sync()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(K):
            one()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / K
        nbytes = eng.opt_span[1] * 4
        allreduce = {"ms": ms, "bytes": nbytes, "algbw_GBps": nbytes / ms / 1e6,
                     "busbw_GBps": nbytes / ms / 1e6 * 2 * (world - 1) / world, "note": "flat gradient bucket alone, back to back; in the step "
                     "the texture / normal-map part runs under the mesh + hand-layer backward"}
    losses = eng.losses()
    finite = all(np.isfinite(v) for v in losses.values())
    consistent = None
    if world > 1 or force_dist:
        # data-parallel invariant (outside the timed region): every rank holds the same parameters after the same all-reduced updates
        cs = eng.p_buf[:eng.opt_span[0] + eng.opt_span[1]].double().abs().sum().reshape(1)
        hi, lo = cs.clone(), cs.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        consistent = bool((hi == lo).item())
    frames = world * eng.B * args.steps
    out = {"metric": "render+loss+backward+Adam frames/sec, 512x512 MANO hand mesh", "value": frames / dt, "unit": "frames/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "C3/C4: 256-frame 512x512 sequence, subdivided MANO hand 3093v/6152f/3327uv, self-shadow, "
                                  "coarse+appearance terms (displacement+texture stage, VGG excluded), dense Adam",
                      "frames_per_gpu_per_step": eng.B, "global_batch": world * eng.B, "sequence_frames": eng.T,
                      "parallelism": f"dp{world} (frames sharded; flat gradient bucket of {eng.opt_span[1] * 4} B all-reduced over RCCL: the texture/normal-map "
                                     f"part overlapped with the mesh backward, the remainder before Adam)",
                      "hipgraph": bool(eng._graphs),
                      "collective": ("harp_allreduce_flat (RCCL on the step's streams, captured into the hipGraph)" if comm is not None else
                                     ("torch.distributed.all_reduce, eager steps" if (world > 1 or force_dist) else None)),
                      "rendered_image": "not materialised: the shader backward recomputes the colour and forms the photometric L1 and its gradient "
                                        "itself, so the step has no forward shading launch (loss, gradients and the parameter update are its "
                                        "outputs; FitEngine.keep_image=True renders and writes y_pred like the reference)"},
           "losses_finite": finite}
    if graph_fallback is not None:
        out["graph_fallback"] = graph_fallback
    if consistent is not None:
        out["ranks_consistent"] = consistent
    if per_rank_ms is not None:
        out["per_rank_ms_per_step"] = per_rank_ms
        out["allreduce"] = allreduce
    if shared_gpu or backend != "nccl":
        out["invalid_timing"] = f"development run: backend={backend}, all ranks on one GPU={shared_gpu}"
    if rank == 0 and world == 1 and not args.no_roofline:
        a_frame, a_step, parts = algorithmic_bytes(eng)
        kt = kernel_roofline(eng, 4)                    # stand-alone: one stream, the group runs alone (HIP events)
        kt_situ = kernel_roofline(eng, 4, overlap=True)   # eager two-stream re-run, events around the C-ABI call
        # ... and the number profiles/ records: the kernel's average duration INSIDE the replayed graph, from a rocprofv3 --kernel-trace
        # child over this same command (events cannot bracket one node of a replayed graph).  `frac` / `achieved` are computed from it;
        # the event timings stay as secondary keys.  Children run outside the timed region; a profiling child never starts children itself.
        child = os.environ.get("HARP_BENCH_CHILD") == "1"
        prof = None if (args.no_profile or child) else profiled_in_graph()
        kg = {}
        if prof:
            for grp, subs in _GROUP_KERNELS.items():
                hit = [avg for name, (avg, n) in prof.items() if any(t in name for t in subs)]
                if hit and grp in kt:
                    kg[grp] = sum(hit) / 1e3
        timing = kg if all(k in kg for k in kt) else kt_situ
        timing_source = ("rocprofv3 --kernel-trace child over `python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-extras --no-roofline`: average "
                         "duration over the 40 timed graph replays (in the graph, next to the other streams' kernels)" if timing is kg else
                         "HIP events around the C-ABI call in an eager two-stream re-run of the steps (no rocprofv3 child: --no-profile, not on PATH, or failed)")
        # dominant KERNEL of the step by measured time: a group's own big kernel decides (subs[0]), not the sum with its set-up launches —
        # the camera-view group (raster + three set-up kernels, 0.232 ms) would otherwise outrank the shader backward (one kernel, 0.227 ms)
        kmain = {}
        if prof:
            for grp, subs in _GROUP_KERNELS.items():
                hit = [avg for name, (avg, n) in prof.items() if subs[0] in name]
                if hit and grp in timing:
                    kmain[grp] = max(hit) / 1e3
        dom = max(kmain, key=kmain.get) if kmain else max(timing, key=timing.get)
        fused = "harp_shade_fwd" not in kt           # loss-only mode: the photometric L1 is formed inside the shader backward
        geom_pos = parts["V"] * 12 + parts["F"] * 12
        S2 = parts["S2"]
        # algorithmic bytes per launch, SURVEY.md §8(d): rasteriser sub-figure geom_pos + S^2*(4+4+12+4) per K=1 pass (face id, z, bary, dist)
        # + S^2*4 for the fused silhouette's alpha (+ mask in, g_alpha out for the fused L1)
        alg = {"raster_cam_fwd(setup+bin+raster)": (geom_pos + S2 * 24 + S2 * 4 + S2 * 8) * eng.B,
               "raster_light_fwd(setup+bin+raster)": (geom_pos + S2 * 24) * eng.B,
               "harp_shade_fwd": (parts["geom"] + S2 * (4 + (12 if eng.keep_image else 0) + 4 + 12 + 4 + 12)) * eng.B,   # + fused photometric L1: y_true, mask in, g_rgb out; the image itself only if kept
               # (fused-loss mode: no forward launch; the backward pass reads target + mask instead of the gradient image)
               "harp_shade_bwd": (parts["geom"] + parts["V"] * 36 + S2 * (4 + (12 + 4 if fused else 12) + 4 + 4)) * eng.B + 2 * eng.Ht * eng.Wt * 12,
               "harp_silhouette_bwd": (geom_pos + S2 * 8 + parts["V"] * 12) * eng.B,
               "harp_depth_bwd": (geom_pos + S2 * 8 + parts["V"] * 12) * eng.B}
        if "harp_texel_reduce" in kt:
            # the texel records written by the shader backward and read back once (36 B per shaded pixel) + both maps' double accumulators
            # (read + cleared) and gradient images (read + written) in the finish pass
            shaded = int(((eng.s["face_c"][:eng.B] >= 0) & (eng.y_sil_col[eng.tfid[:eng.B].long()] != 0)).sum().item())
            alg["harp_texel_reduce"] = shaded * 36 + 2 * eng.Ht * eng.Wt * 3 * (8 + 4 + 4)
            # (the shader backward's own figure stays SURVEY 8(d)'s — the path's algorithmic bytes; that it now writes 36-byte records instead of
            #  scattering into the two gradient maps is reported next to it, not folded into `frac`)
            records_bytes = shaded * 36
        # ... and the bytes the FUSED rasteriser kernels really have to move (barycentrics / distances are never materialised): their
        # `frac` is against THIS figure; the one against §8(d)'s formula is kept as `frac_survey_8d`
        moved = {"raster_cam_fwd(setup+bin+raster)": (geom_pos + S2 * (4 + 4) + S2 * (4 + 4)) * eng.B,     # face id + alpha out; mask in, g_alpha out
                 "raster_light_fwd(setup+bin+raster)": (geom_pos + S2 * (4 + 4)) * eng.B}                # face id + depth out
        ach = alg[dom] / (timing[dom] * 1e-3) / 1e9
        # HBM traffic per launch: two PMC passes (FETCH_SIZE, WRITE_SIZE) over this command, run now as children when rocprofv3 is there;
        # otherwise the last committed passes (profiles/traffic_latest.json, stamped with the commit they were taken at)
        # (skipped when the kernel-trace child did not come back: a profiler that is broken on this box must not cost the run three time-outs)
        tjson = None if (args.no_profile or child or prof is None) else profiled_traffic()
        if tjson is not None:
            traffic_source = "measured in this run: " + tjson.get("_source", "")
        else:
            tjson = {}
            tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
            if os.path.exists(tpath):
                tjson = json.load(open(tpath))
            traffic_source = ("NOT measured in this run: per-launch FETCH_SIZE / WRITE_SIZE of the last committed rocprofv3 passes over this command, "
                              f"profiles/traffic_latest.json (taken at commit {tjson.get('_head', 'unknown')})")
        traffic = tjson.get(dom)
        # the same figures for every timed kernel group (north_star asks for the rasteriser's fraction explicitly)
        per_kernel = {}
        for k in kt:
            byts = moved.get(k, alg[k])
            e = {"ms": timing[k], "in_graph_ms": kg.get(k), "standalone_event_ms": kt[k], "in_situ_event_ms": kt_situ.get(k),
                 "bytes": byts, "bytes_kind": ("moved by the fused kernel (no bary / dist images)" if k in moved else "algorithmic, SURVEY.md 8(d)"),
                 "achieved_GBps": byts / (timing[k] * 1e-3) / 1e9, "frac": byts / (timing[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": tjson.get(k)}
            if k in moved:
                e["bytes_survey_8d"] = alg[k]
                e["frac_survey_8d"] = alg[k] / (timing[k] * 1e-3) / 1e9 / HBM_PEAK_GBS
            if tjson.get(k):
                e["frac_of_measured_traffic"] = tjson[k] / (timing[k] * 1e-3) / 1e9 / HBM_PEAK_GBS
            per_kernel[k] = e
        # the roof that binds: none of these kernels is near the HBM roof (traffic <= the algorithmic bytes at 5 - 12 % of peak).  issue_frac =
        # VALU wave-instructions per launch (SQ_INSTS_VALU, one PMC child) x the measured 2.5 cycles per instruction / (1024 SIMDs x 2.4 GHz x
        # the kernel's duration): the share of the chip's vector issue slots the launch fills; atomic_frac (shader backward) = its ~17 M
        # memory-side float atomics (11.8 M texel + 4 M vertex + 2.3 M shadow-window, per B = 32 launch) at the measured 330 G / s
        valu = None if (args.no_profile or child or prof is None) else profiled_issue()
        if valu:
            for k, subs in _GROUP_KERNELS.items():
                main = subs[0]                                    # the group's big kernel (set-up kernels are small and shared by both views)
                hit = [v for n, v in valu.items() if main in n]
                if hit and k in per_kernel:
                    wi = max(hit)
                    per_kernel[k]["valu_wave_instructions"] = wi
                    per_kernel[k]["issue_frac"] = wi * VALU_ISSUE_CYCLES / (SIMDS * CLOCK_HZ * timing[k] * 1e-3)
        if "harp_shade_bwd" in per_kernel and "harp_texel_reduce" in kt:
            per_kernel["harp_shade_bwd"]["texel_record_bytes_written"] = records_bytes
        if "harp_shade_bwd" in per_kernel:
            # (table form: 11.8 M texel + 4 M vertex + 2.3 M shadow-window atomics per B = 32 launch; with texel records the texel part leaves
            #  the kernel: harp_texel_reduce adds ~1.5 M double atomics of its own)
            n_at = (6.3e6 if "harp_texel_reduce" in kt else 17.0e6) * eng.B / 32.0
            per_kernel["harp_shade_bwd"]["memory_atomics_model"] = n_at
            per_kernel["harp_shade_bwd"]["atomic_frac"] = n_at / ATOMIC_RATE / (timing["harp_shade_bwd"] * 1e-3)
        step_s = dt / args.steps
        step_bytes = a_frame * eng.B + a_step
        # loss-only mode writes no y_pred and reads no gradient image back: 2 * S^2 * 12 B per frame less than §8(d)'s A_frame
        step_bytes_lean = step_bytes - 2 * S2 * 12 * eng.B
        out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                           "traffic": traffic, "traffic_source": traffic_source,
                           "algorithmic_bytes_per_launch": alg[dom], "avg_ms": timing[dom], "timing_source": timing_source,
                           "standalone_event_ms": kt[dom], "in_situ_event_ms": kt_situ.get(dom),
                           "standalone_frac": alg[dom] / (kt[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "kernel_ms": timing, "per_kernel": per_kernel, "step_algorithmic_bytes": step_bytes,
                           "step_frac_of_hbm_roofline": step_bytes / step_s / 1e9 / HBM_PEAK_GBS,
                           "step_algorithmic_bytes_loss_only_mode": step_bytes_lean,
                           "step_frac_of_hbm_roofline_loss_only_bytes": step_bytes_lean / step_s / 1e9 / HBM_PEAK_GBS,
                           "issue_frac": per_kernel[dom].get("issue_frac"), "atomic_frac": per_kernel[dom].get("atomic_frac"),
                           "binding_roof": ("not HBM: the kernel moves <= its algorithmic bytes at `frac` of the HBM peak; `issue_frac` of the chip's VALU issue "
                                            "slots (2.5 cycles per wave64 instruction, measured) and `atomic_frac` of the memory-side atomic rate are filled "
                                            "— the remainder is latency between tile-round barriers / LDS-table phases at 4 waves per SIMD (DESIGN.md 6)")}
        if prof:
            short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            out["roofline"]["kernels_in_graph_us"] = {short(n): round(a, 2) for n, (a, c) in sorted(prof.items(), key=lambda kv: -kv[1][0] * kv[1][1])
                                                      if c >= 40 and "at::native" not in n and "Cijk" not in n}
        # the like-for-like step that materialises y_pred as the reference does, next to the headline (always measured: 40 replays)
        eng.keep_image = True
        ki = _graph_rate(eng, 40, 6)
        eng.keep_image = False
        out["value_keep_image"] = ki["frames_per_s"]
        out["ms_per_step_keep_image"] = ki["ms_per_step"]
        out["roofline"]["step_frac_of_hbm_roofline_keep_image"] = step_bytes / (ki["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        if not args.no_extras:
            out["extras"] = extra_rates(eng, device, vgg_weights=None if args.vgg_weights == "none" else args.vgg_weights)
            # capture-to-capture spread of the headline step (hipGraph re-draws its stream assignment at every capture: 2 - 3 %): 5 fresh
            # captures x 200 replays; gains below this band are not gains
            caps = []
            for _ in range(5):
                eng._graphs = {}
                caps.append(_graph_rate(eng, 200, 10)["ms_per_step"])
            out["extras"]["headline_step_ms_over_5_captures"] = {"min": min(caps), "median": float(np.median(caps)), "max": max(caps), "replays_each": 200}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        try:                                             # (RCCL prints its version banner through C stdio: keep the JSON line the LAST line of stdout)
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:                                # noqa: BLE001
            pass
        print(json.dumps(out), flush=True)
    if comm is not None:
        torch.cuda.synchronize()
        comm.destroy()
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
