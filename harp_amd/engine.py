"""Fused render-and-compare step: the loop body of the reference's `optimize_hand_sequence`
(optimize_sequence.py:446-579) as one pre-planned sequence of HIP kernel launches over pre-allocated HBM buffers,
replayable as a hipGraph, with one RCCL all-reduce of the flat gradient arena between backward and Adam.

What the reference does per step with ~600 torch/PyTorch3D kernel launches, >= 8 host syncs and CPU-resident
parameters, this does with 28 launches (hand mesh, texel records; 24 in the table form; 28 + 4 for the SMPL-X arm), no host sync and everything resident:

  hand_front (schedule row, frame set-up, LBS, subdivide, normals + displace, normals, both projections, light camera; arm: 5 launches) ->
  raster(cam, K=1 + soft silhouette + its L1) || raster(light, K=1) || parameter / mesh regularisers ->
  shade_bwd (recomputes the colour, forms the photometric L1, writes y_pred when asked) || silhouette_bwd -> depth_bwd ->
  mesh chain + hand / arm layer backward on four workgroups per frame (6 / 7 launches) -> [all-reduce] -> Adam

(the building blocks behind the fused launches — frame_setup, LBS, subdivide, normals, project, centroid, light_setup, shade and
their backward passes — are separate C-ABI entry points and stay reachable through the `fused_*` switches; tests compare the two).

Parameters live in ONE flat fp32 arena (and one gradient / exp_avg / exp_avg_sq arena of the same layout); the
reference's parameter dict (optimize_sequence.py:181-250) is exposed as views into it (`params`).
Multi-GPU: frames are sharded over ranks, every rank holds the full arena; gradients are summed with one all_reduce
and scaled by 1/world inside the Adam kernel (SURVEY.md §5 "Data-parallel semantics").
"""
import ctypes
import os
import math

import numpy as np
import torch

from . import _lib, ops
from .manopth.manolayer import ManoDeviceModel

LOSS_NAMES = ["silhouette", "kps_anchor", "vert_disp_reg", "laplacian", "normal", "arap", "photo", "albedo", "normal_reg"]
LOSS_WEIGHTS = {"silhouette": 7.0, "kps_anchor": 10.0, "vert_disp_reg": 2.0, "laplacian": 4.0, "normal": 0.1, "arap": 0.2,
                "photo": 1.0, "albedo": 0.5, "normal_reg": 0.1}                     # optimize_sequence.py:411-422 (vgg: §8f "next")
BG_COLOR = (1.0, 1.0, 1.0)          # background of the shading pass (renderer_helper.py BlendParams default)
COARSE_TERMS = LOSS_NAMES[:6]
APP_TERMS = LOSS_NAMES[6:]


class _Arena:
    """Flat fp32 buffer with named, 64-float aligned segments."""

    def __init__(self, spec, device):
        self.offsets, off = {}, 0
        for name, shape in spec:
            n = int(np.prod(shape)) if len(shape) else 1
            self.offsets[name] = (off, n, tuple(shape))
            off += (n + 63) // 64 * 64
        self.size = off
        self.device = device

    def alloc(self):
        return torch.zeros(self.size, dtype=torch.float32, device=self.device)

    def view(self, buf, name):
        off, n, shape = self.offsets[name]
        return buf[off:off + n].view(shape)

    def span(self, first, last):
        o0 = self.offsets[first][0]
        o1, n1, _ = self.offsets[last]
        return o0, (o1 + n1 + 63) // 64 * 64 - o0


class FitEngine:
    """One rank's share of a HARP fitting job.

    model: MANO-shaped dict (numpy) — v_template, shapedirs, posedirs, J_regressor, weights, hands_mean.
    topo: harp_amd.synth.build_topology(...) dict; verts_uvs/faces_uvs/uv_mask: template UV data.
    input_params: dict of (T,.) tensors as `init_params` consumes (pose, rot, trans, shape, cam, joints).
    """

    def __init__(self, model, topo, verts_uvs, faces_uvs, uv_mask, input_params, img_size, focal_length, batch_size,
                 device="cuda", self_shadow=True, share_light_position=True, tex_size=512, rank=0, world_size=1, seed=0,
                 use_arm=False, opt_arm_pose=False):
        self.dev = torch.device(device)
        self.S, self.focal, self.B = int(img_size), float(focal_length), int(batch_size)
        self.self_shadow, self.share_light = bool(self_shadow), bool(share_light_position)
        self.rank, self.world = rank, world_size
        self.topo = ops.DeviceTopology(topo, verts_uvs, faces_uvs, self.dev)
        self.use_arm, self.opt_arm_pose = bool(use_arm), bool(opt_arm_pose)
        if self.use_arm:                                                              # SMPL-X right arm (config use_arm, utils/config_utils.py:6)
            from .hand_models_harp.body_models import TreeDeviceModel
            self.dm = TreeDeviceModel(model, self.dev)
            self.n_joints, self.pose_stride, self.n_betas = 22, 51, self.dm.NB
        else:
            self.dm = ManoDeviceModel(model, self.dev)
            self.n_joints, self.pose_stride, self.n_betas = 21, 48, 10
        T = input_params["pose"].shape[0]
        self.T, V = T, self.topo.V
        self.Ht = self.Wt = tex_size
        # ---- parameter arena: [coarse group | appearance group | not optimised]  (optimize_sequence.py:253-310)
        spec = [("pose", (T, 45)), ("cam", (T, 3)), ("verts_disps", (V, 1)), ("shape", (10,)), ("rot", (T, 3)), ("wrist_pose", (T, 3)),
                ("light_positions", (T, 3)), ("amb_ratio", ()), ("texture", (1, tex_size, tex_size, 3)), ("normal_map", (1, tex_size, tex_size, 3)),
                ("trans", (T, 3))]
        self.arena = _Arena(spec, self.dev)
        self.p_buf, self.m_buf, self.v_buf = (self.arena.alloc() for _ in range(3))
        # everything that is zeroed at the start of a step lives in ONE slab (a single fill kernel): the lane's per-frame gradient
        # scratch, the normal-map gradient, the gradient arena and the loss vector
        self.fid = torch.zeros(self.B, dtype=torch.int32, device=self.dev)
        self.tfid = torch.zeros(self.B, dtype=torch.int32, device=self.dev)
        self.nmap_n = torch.empty(tex_size, tex_size, 3, dtype=torch.float32, device=self.dev)
        self.texnm = torch.empty(tex_size, tex_size, 8, dtype=torch.float32, device=self.dev)      # interleaved albedo + normal map
        self._main = self._alloc_lane(self.B, 0, extra=[("g_nmap_n", (tex_size, tex_size, 3)), ("g_buf", (self.arena.size,)), ("loss_vec", (16,))])
        self.g_buf, self.g_nmap_n, self.loss_vec = (self._main["s"][k] for k in ("g_buf", "g_nmap_n", "loss_vec"))
        self.params = {k: self.arena.view(self.p_buf, k) for k, _ in spec}
        self.grads = {k: self.arena.view(self.g_buf, k) for k, _ in spec}
        # rot / wrist_pose join the coarse group only under use_arm & opt_arm_pose (optimize_sequence.py:264-268, 279-284)
        self.coarse_span = self.arena.span("pose", "wrist_pose" if (self.use_arm and self.opt_arm_pose) else "shape")
        self.app_span = self.arena.span("light_positions", "normal_map")
        self.opt_span = (self.coarse_span[0], self.app_span[0] + self.app_span[1] - self.coarse_span[0])
        with torch.no_grad():                                                         # init_params (optimize_sequence.py:181-250)
            for k in ("pose", "rot", "trans", "cam"):
                self.params[k].copy_(input_params[k].to(self.dev))
            self.params["shape"].copy_(input_params["shape"].mean(0).to(self.dev))
            self.params["texture"].copy_((torch.tensor([232, 190, 172]).repeat(1, tex_size, tex_size, 1) / 255.).to(self.dev))
            self.params["normal_map"].copy_(torch.tensor([0.0, 0.0, 1.0]).repeat(1, tex_size, tex_size, 1).to(self.dev))
            self.params["light_positions"].copy_(torch.tensor(((-0.5, -0.5, -0.5),)).repeat(T, 1).to(self.dev))
            self.params["amb_ratio"].fill_(0.4)
        self.uv_mask = torch.as_tensor(np.asarray(uv_mask), dtype=torch.float32).to(self.dev).contiguous() if uv_mask is not None else None
        self.init_joints = input_params["joints"].to(self.dev).float().contiguous() if "joints" in input_params else None
        # ---- frame tables struct
        t = _lib.FrameTables()
        for k in ("pose", "rot", "trans", "cam", "shape", "light_positions", "amb_ratio"):
            setattr(t, k, _lib.ptr(self.params[k]))
            setattr(t, "g_" + k, _lib.ptr(self.grads[k]))
        t.share_light = int(self.share_light)
        if self.use_arm:
            t.wrist_pose, t.g_wrist_pose = _lib.ptr(self.params["wrist_pose"]), _lib.ptr(self.grads["wrist_pose"])
        t.n_betas_out = self.n_betas
        self.tables = t
        # ---- Adam hyper-parameters on the device (coarse lr 1e-3, appearance lr 1e-2; torch defaults otherwise)
        self.hyper_np = np.zeros(2, dtype=[("lr", "f4"), ("beta1", "f4"), ("beta2", "f4"), ("eps", "f4"), ("grad_scale", "f4"),
                                           ("step", "i4"), ("step_size", "f4"), ("inv_sqrt_bc2", "f4")])
        self.hyper_np["lr"] = [1e-3, 1e-2]
        self.hyper_np["beta1"], self.hyper_np["beta2"], self.hyper_np["eps"] = 0.9, 0.999, 1e-8
        self.hyper_np["grad_scale"] = 1.0 / world_size
        self.hyper = torch.from_numpy(self.hyper_np.view(np.uint8).copy()).to(self.dev)
        self._hyper_stride = self.hyper_np.dtype.itemsize
        # ---- targets (set by set_targets) and per-step scratch
        self.y_true = self.y_sil = self.y_sil_col = None
        self.bg_sil = self.bg_photo = None
        self.target_offset = 0
        self.w_vec = torch.zeros(16, dtype=torch.float32, device=self.dev)
        self._main["loss_vec"], self._main["w_vec"] = self.loss_vec, self.w_vec
        # accumulate_loss: loss_total += sum_k w_k loss_k after every step — the reference's per-step `sum_loss` added up over an epoch
        # (optimize_sequence.py:553-559, :581) — by hand_back itself in a folded step, by two small torch kernels otherwise
        self.accumulate_loss = False
        self.loss_total = torch.zeros(1, dtype=torch.float32, device=self.dev)
        self.w_total = torch.zeros(16, dtype=torch.float32, device=self.dev)
        self.loss_acc = torch.zeros(16, dtype=torch.float32, device=self.dev)      # fold_step: the terms accumulate here, hand_back moves them to loss_vec and clears
        self._main["owns_shared"] = True               # its zero slab also covers g_buf / g_nmap_n / loss_vec
        self._activate(self._main)
        self.dist_albedo = torch.zeros(tex_size, tex_size, 2, dtype=torch.int32, device=self.dev)
        self.dist_normal = torch.zeros(tex_size, tex_size, 2, dtype=torch.int32, device=self.dev)
        self.seed = int(seed) & 0x7FFFFFFF              # SAME seed on every rank
        self.draw_counter = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self.ref_verts = None
        self._graphs = {}
        # per-frame fused mesh chain (csrc/chain.hip): 22 launches -> 2; needs the frame's mesh to fit its LDS staging
        self.fused_chain = self.topo.V <= _lib.lib().harp_mesh_chain_max_vertices()
        # MANO path: frame set-up + hand layer + mesh chain + rasteriser set-up of both views as ONE launch (csrc/hand_front.hip)
        # SMPL-X arm path: the same as three + four launches around the shared MFMA contractions (csrc/arm_front.hip)
        self.fused_front = self.fused_chain and (self.n_joints == 21 or self.use_arm)
        if self.use_arm:
            self._weights_T = self.dm.weights.t().contiguous()                       # (NJ, NV): one coalesced row per joint for the per-frame kernels
        self.fused_back = True           # ... and the backward tail as three launches instead of six (csrc/hand_back.hip)
        # the front on FOUR workgroups per frame (csrc/chain_wide.hip, hand_front_wide_kernel): a quarter of the vertices per workgroup, one
        # pass per stage, kernel boundaries where the parts meet — three launches of 4 B workgroups instead of one of B
        # Measured (tools/dev/bench_ab.sh, fresh processes = the burst regime bench.py reports; tools/dev/gpu_wide_ab.py, sustained): the wide TAIL
        # wins everywhere (hand -13 ... -16 us / step, arm -25); the wide FRONT wins on the arm (-10 ... -35 us) and in sustained runs of the hand
        # path (-17 us), but LOSES on the hand path in a fresh process (+12 ... +20 us: three dependent nodes instead of one in front of a
        # 0.65-ms step whose second stream is the longer branch of the fork) — off there.
        wide_ok = self.fused_front and (self.topo.V + 3) // 4 <= 1024
        self.wide_front = wide_ok and self.use_arm
        self.hybrid_front = False            # hand path: hand layer on four workgroups per frame + one-workgroup mesh chain (two launches)
        self.front_auto = wide_ok and not self.use_arm   # hand path: the form of the front is chosen per stage (see _mesh_forward); False: wide_front / hybrid_front as set
        self.wide_back = wide_ok             # the tail: mesh-chain backward + per-vertex hand / arm layer backward on four workgroups per frame
        # ---- switches (all on in production; tools/dev and bench.py flip some of them to measure their effect)
        self.overlap = True              # second HIP stream (light view, silhouette backward, parameter-only terms); False: one stream
        self.early_terms = True          # parameter-only terms / mesh regularisers scheduled on the second stream
        self.packed_texels = True        # shaders read the interleaved albedo + normal-map array (harp_pack_texels)
        self.auto_draw = True            # draw fresh texture-regulariser offsets every step (False: the caller draws)
        self.overlap_allreduce = True    # N > 1: all-reduce of the map gradients overlapped with the mesh / LBS backward
        self.graph_collectives = False   # N > 1 over torch.distributed (no RcclComm): capture its all-reduce into the step graph (opt-in)
        self.comm = None                 # harp_amd.dist.RcclComm: direct RCCL all-reduce on the step's stream (graph node by default), set_comm()
        self.force_allreduce = False     # run the N > 1 code path on a single rank (tests, bench HARP_FORCE_DIST)
        self.fused_loss = True           # loss-only mode: photometric L1 formed inside the shader backward (no forward shading launch)
        self.graph_order = True
        self.mesh_third = False          # key-point / mesh regularisers on a third stream (their own graph branch) instead of in front of the light view.  Round 3: 0.755 vs 0.766 ms / step with the third stream (three launches incl. their clear).  Round 4, with ONE launch for the terms, the clear inside hand_front and the texture regularisers throttled: 0.6665 vs 0.6695 without it (the shader backward joins one stream instead of two; the light view has the slack), C5 1.740 vs 1.752 — off
        self.keep_depth = True           # light-view depth map kept across steps (harp_rasterize_fwd_keep): empty super-tiles are filled with -1 once, not every step (25 MB)
        self.consume_gzl = True          # the depth backward clears the shadow-map gradient entries it consumes: no per-step clear of that image (33.5 MB at B = 32, 512^2)
        self.fused_keep = True           # keep_image with the fused loss: the shader backward also writes y_pred (no forward shading launch either)
        self.keep_image = True           # shader forward writes the rendered image s["rgb"] (False: loss + gradient only)
        self.mesh_terms_late = False     # key-point / mesh terms on the second stream behind the silhouette backward (beside the shader backward)
        self.sil_late = False            # silhouette backward after the shader backward instead of beside it (measured: see profiles/r05_wide_ab.txt)
        self.paired_setup = False        # rasteriser set-up of both views as three launches on the main stream (harp_raster_setup_pair; the light raster no longer waits for three set-up launches of its own on the second stream).  Measured, same box: hand +7 us / step (B = 32), +5 (B = 18), arm +15; together with wide_front -3 ... +5: off (profiles/r05_wide_ab.txt)
        self.late_texture_terms = False  # texture regularisers behind the light view on the second stream (see forward_backward)
        self.mesh_terms_first = True     # key-point term + mesh regularisers run before the light raster (under the raster set-up) instead of after it
        self.fused_bwd = False           # shading + silhouette backward in ONE launch (harp_shade_sil_bwd): correct, measured SLOWER (1.05 vs 0.93 ms: the rasteriser tiles inherit 168 VGPRs / 3 waves per SIMD)
        self.tail_side = False           # normal-map chain rule (+ early all-reduce) on the second stream: measured SLOWER (0.960 vs 0.948 ms: the extra cross-stream edge costs more than the 5-us kernel it moves)
        self.camera_first = True        # enqueue the camera-view raster chain (the longer one) before the light-view chain: +0.75 %
        # the shader backward flags the light-view tiles it adds a shadow-tap gradient to and the depth backward reads only those (42 % of the
        # tiles it visits at 512^2, 51 % at 1024^2 on the arm).  The flag is one more dependent load per tile: at 512^2, where a workgroup of
        # the depth backward walks <= 2 tiles, it costs what it saves (0.704 vs 0.701 ms / step); at 1024^2 (8 tiles per workgroup) it wins
        # (1.777 vs 1.788 ms / step on the arm) — on from 1024 px.
        self.zl_tile_flags = self.S >= 1024
        # the shader backward hands the texture / normal-map gradients on as one record per shaded pixel, binned by 32x32-texel UV tile, and
        # harp_texel_reduce (csrc/texel_reduce.hip) adds them up on a branch of its own that joins in front of Adam — beside the mesh / hand
        # backward tail instead of inside the shader backward (its LDS texel table, flush and ~12 M memory atomics per launch are gone)
        self.texel_records = _lib.lib().harp_texel_bins(self.Ht, self.Wt) <= 1024 and world_size == 1      # (N > 1: the table form — the map gradients are final ~60 us earlier, which is what their early all-reduce overlaps with; switchable)
        self.trec_cap_div = 32           # capacity of a tile's record list = B * S * S / this (>= 65536): 3.5x the fullest list of the bench scenes; a full list falls back to memory atomics
        self.trec_cap_min = 65536
        self._trec = self._tacc = None
        self._maps_pending = None
        self.vert9 = True                # the shader backward's vertex gradients as ONE interleaved (B,V,9) buffer (a 36-byte run per vertex and wave instead of three 12-byte runs: a third of the memory-atomic lines), unpacked into the three arrays by extra workgroups of the depth backward's launch
        self.vgg_streams = 2             # perceptual term: the batch in this many parts on as many streams (harp_vgg16_term_args.side_streams; 1 - 4)
        self.split_adam = True           # with the texel records: the maps' Adam update on the second stream behind harp_texel_finish, the step's last launch only for the small parameters
        self.fused_sil_bwd = False       # the silhouette backward inside the camera-view raster launch (harp_rasterize_l1_fwd_bwd) instead of a launch of its own beside the shader backward.  Correct (tests) and measured SLOWER: the shader backward gains 32 us without its neighbour (230 -> 198 in the graph), the camera raster pays 56 (198 -> 254: 94 VGPRs / 26 KB of LDS = 5 waves per SIMD instead of 7, and the rim walk is ~25 us of VALU work wherever it runs): step 0.665 vs 0.638 ms (profiles/r06_ab_record.txt)
        self.lean_app_stage = False      # appearance-only stage without the geometry gradients nothing reads (set by optimize_hand_sequence; off by default: g_buf then holds what autograd would)
        self._lean_now = False
        self.sil_only_raster = True      # geometry-only steps without a kept image: the camera raster forms no nearest-face ids (harp_rasterize_l1_fwd with face_id == NULL)
        self.fold_step = True            # scheduled steps: the batch row is fetched by hand_front itself, the loss vector / schedule row / draw counter are turned over by hand_back, the slab clear + Adam tick + offset draw are ONE launch (harp_step_frame, harp_step_prologue): 31 -> 23 kernels per step, no schedule kernel in front of the hand layer
        self.fused_terms = True          # normalise + pack, the four parameter-only regularisers, key-point + mesh terms, depth backward + normal-map chain rule: one launch each (were 2 + 4 + 2 + 2)
        self.frozen = ()                 # parameters kept out of the optimiser groups (known_appearance)
        self.disabled_terms = frozenset()   # loss terms left out of the objective altogether (set_disabled_terms)
        self.schedule = self.tschedule = None
        self._stage = None
        self._early_work = None
        self._loss_cleared = False
        self.perceptual = None           # optional VGG feature term of the appearance stage (set_perceptual)
        self.graph_perceptual = True
        # HARP_ENG="switch=0,other=1": schedule switches of this engine from the environment (A/B runs of bench.py and the tools; the
        # switches are all result-neutral, tests/test_gpu_parity.py::test_schedule_switches_give_the_default_schedules_result)
        for kv in filter(None, os.environ.get("HARP_ENG", "").split(",")):
            k, v = kv.split("=")
            if not hasattr(self, k):
                raise ValueError(f"HARP_ENG: no engine switch {k!r}")
            setattr(self, k, type(getattr(self, k))(int(v)))
            if k in ("wide_front", "hybrid_front"):
                self.front_auto = False                  # an explicit form is an explicit form
        self.compute_reference_mesh()

    # ------------------------------------------------------------------------------------------------
    def _alloc_lane(self, B, lo, extra=()):
        """scratch + bookkeeping for B frames starting at position `lo` of the step's batch"""
        dev, V, S = self.dev, self.topo.V, self.S
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        L = _lib.lib()
        s = {}
        s["pose48"], s["betas"], s["trans_b"] = f(B, self.pose_stride), f(B, self.n_betas), f(B, 3)
        s["cam_R"], s["cam_T"], s["light_pos"], s["colors"] = f(B, 9), f(B, 3), f(B, 3), f(9)
        V0, NJo = self.topo.V0, self.n_joints
        s["lbs_ws"] = f(L.harp_lbs_tree_ws_floats(ctypes.byref(self.dm.struct), B) if self.use_arm else L.harp_lbs_mano_ws_floats(B))
        s["verts_mm"], s["joints_mm"], s["joints_m"] = f(B, V0, 3), f(B, NJo, 3), f(B, NJo, 3)
        s["chain_parts"] = f(L.harp_mesh_chain_wide_ws_floats(B, V))
        s["vs"], s["n1"], s["il1"], s["vd"], s["n2"], s["il2"] = f(B, V, 3), f(B, V, 3), f(B, V), f(B, V, 3), f(B, V, 3), f(B, V)
        s["ndc_c"], s["ndc_l"], s["centroid"], s["light_R"], s["light_T"] = f(B, V, 3), f(B, V, 3), f(B, 3), f(B, 9), f(B, 3)
        s["ws_c"] = ops.rasterize_workspace(B, self.topo.F, S, dev)
        s["ws_l"] = ops.rasterize_workspace(B, self.topo.F, S, dev)
        s["face_c"] = torch.empty(B, S, S, dtype=torch.int32, device=dev)
        s["face_l"] = torch.empty(B, S, S, dtype=torch.int32, device=dev)
        s["alpha"], s["zl"], s["rgb"] = f(B, S, S), f(B, S, S), f(B, S, S, 3)
        s["g_v9"] = torch.zeros(B, V, 9, dtype=torch.float32, device=dev)       # harp_shade_args.g_vert9: all-zero between steps (its unpack clears it)
        s["zl_tiles"] = torch.zeros(B * ((S + 15) // 16) ** 2, dtype=torch.uint8, device=dev)     # harp_shade_args.g_zl_tiles: all-zero between steps
        s["zl_state"] = torch.zeros(B * ((S + 63) // 64) ** 2, dtype=torch.int32, device=dev)     # harp_rasterize_fwd_keep: which super-tiles of zl are all -1
        s["nmap_n"] = self.nmap_n
        # gradients (zeroed every step in ONE memset: they are carved from one flat buffer)
        gspec = [("g_alpha", (B, S, S)), ("g_rgb", (B, S, S, 3)), ("g_zl", (B, S, S)), ("g_vd", (B, V, 3)), ("g_joints_m", (B, NJo, 3)), ("g_n2", (B, V, 3)),
                 ("g_ndc_c", (B, V, 3)), ("g_ndc_l", (B, V, 3)), ("g_n1", (B, V, 3)), ("g_vs", (B, V, 3)), ("g_tmp", (B, V, 3)),
                 ("g_v0", (B, V0, 3)), ("g_joints_mm", (B, NJo, 3)), ("g_light_pos", (B, 3)), ("g_colors", (9,)),
                 ("g_light_R", (B, 9)), ("g_light_T", (B, 3)), ("g_cam_R", (B, 9)), ("g_cam_T", (B, 3)), ("g_centroid", (B, 3)),
                 ("g_pose48", (B, self.pose_stride)), ("g_betas", (B, self.n_betas)), ("g_trans_b", (B, 3))] + list(extra)
        garena = _Arena(gspec, dev)
        gs_buf = garena.alloc()
        for k, _ in gspec:
            s[k] = garena.view(gs_buf, k)
        if "g_nmap_n" not in s:
            s["g_nmap_n"] = self.g_nmap_n
        # g_alpha and g_rgb (the first two segments, 4/5 of the slab) are fully overwritten by harp_image_l1: only the rest is zeroed
        # ... and of the rest, g_zl (B*S*S floats, 9/10 of it) is first touched by the shader backward: it is cleared on the second
        # stream, off the head of the step; the small remainder is cleared first thing on the main stream
        # ... and g_vd / g_joints_m, the two the key-point / mesh terms accumulate into, are a segment of their own: those terms run on a
        # third stream, which clears the segment itself instead of depending on the clear of another branch
        return dict(s=s, gs_zero=gs_buf[garena.offsets["g_n2"][0]:], gs_mesh=gs_buf[garena.offsets["g_vd"][0]:garena.offsets["g_n2"][0]], gs_zero_late=s["g_zl"], B=B, lo=lo, fid=self.fid[lo:lo + B], tfid=self.tfid[lo:lo + B],
                    loss_vec=torch.zeros(16, dtype=torch.float32, device=dev), w_vec=torch.zeros(16, dtype=torch.float32, device=dev),
                    stream=None, side=None)

    # `consume_gzl` / `keep_depth` carry invariants ACROSS steps (g_zl is all-zero between steps because its consumer clears it; zl_state
    # says which super-tiles of the kept light depth map are all -1).  A step with the switch off breaks the invariant (g_zl stays dirty, the
    # plain rasteriser fills super-tiles the state calls empty), so a flip re-establishes it before the next step.
    def _get_consume_gzl(self):
        return self._consume_gzl

    def _set_consume_gzl(self, v):
        if getattr(self, "_consume_gzl", bool(v)) != bool(v):
            self._shadow_state_stale = True
        self._consume_gzl = bool(v)

    def _get_keep_depth(self):
        return self._keep_depth

    def _set_keep_depth(self, v):
        if getattr(self, "_keep_depth", bool(v)) != bool(v):
            self._shadow_state_stale = True
        self._keep_depth = bool(v)

    consume_gzl = property(_get_consume_gzl, _set_consume_gzl)
    keep_depth = property(_get_keep_depth, _set_keep_depth)

    def _reset_shadow_state(self):
        for lane in (self._main,):
            for k in ("g_zl", "zl_tiles", "zl_state"):
                lane["s"][k].zero_()
        self._shadow_state_stale = False

    def _activate(self, lane):
        """point the step code at one lane's buffers (host-side bookkeeping only)"""
        self.s, self.gs_zero, self.gs_mesh, self.gs_zero_late, self._lane = lane["s"], lane["gs_zero"], lane["gs_mesh"], lane["gs_zero_late"], lane

    def set_targets(self, y_true, y_sil, y_sil_col, frame_offset=0):
        """(Tl,S,S,3), (Tl,S,S), (Tl,S,S) fp32 for this rank's frames [frame_offset, frame_offset+Tl): kept resident in HBM
        (the reference re-reads them from 20 DataLoader workers + H2D every step, optimize_sequence.py:446-450)."""
        self.y_true = y_true.to(self.dev).float().contiguous()
        self.y_sil = y_sil.to(self.dev).float().contiguous()
        self.y_sil_col = y_sil_col.to(self.dev).float().contiguous()
        self.target_offset = int(frame_offset)
        self._graphs = {}                               # captured graphs hold the raw pointers of the previous target buffers
        # the targets are static during a fit: what an un-rendered 64x64 super-tile contributes to the two image terms is a constant per
        # (target frame, super-tile) — sum of y_sil for the silhouette L1 (alpha = 0), sum of |bg - y| * mask for the photometric L1 —
        # tabulated once here; the loss-only kernels (keep_image = False) look it up instead of reading 3/4 of the targets every step
        T, S, nsx = self.y_sil.shape[0], self.S, (self.S + 63) // 64
        pad = nsx * 64 - S
        def tile_sums(img):                                                          # (T,S,S) -> (T, nsx*nsx), super-tile st = sy * nsx + sx
            x = torch.nn.functional.pad(img.double(), (0, pad, 0, pad))
            return x.reshape(T, nsx, 64, nsx, 64).sum((2, 4)).reshape(T, nsx * nsx).float().contiguous()
        self.bg_sil = tile_sums(self.y_sil)
        bg = torch.tensor(BG_COLOR, dtype=torch.float32, device=self.dev)
        m = self.y_sil_col.unsqueeze(-1)
        self.bg_photo = tile_sums((bg * m - self.y_true * m).abs().sum(-1))
        if getattr(self, "perceptual", None) is not None:                            # cached target features belong to the old targets
            self.set_perceptual(self._vgg_module, self.perceptual_weight, cache_bytes=self._vgg_cache_bytes, precision=self._vgg_precision,
                                bounded=self._vgg_bounded)

    # ------------------------------------------------------------------------------------------------
    def _ck(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed with status {rc}")

    def _chain_struct(self, B, shadow, has_normal_grad):
        """harp_mesh_chain over the active lane's scratch (fused per-frame mesh chain, csrc/chain.hip)"""
        s, p, tp = self.s, _lib.ptr, self.topo
        c = _lib.MeshChain()
        for k, t in (("edges0", tp.edges0), ("vf_off", tp.vf_off), ("vf_tri", tp.vf_tri), ("sub_off", tp.sub_off),
                     ("sub_idx", tp.sub_idx), ("disp", self.params["verts_disps"]), ("verts_mm", s["verts_mm"]), ("joints_mm", s["joints_mm"]),
                     ("cam_R", s["cam_R"]), ("cam_T", s["cam_T"]), ("light_pos", s["light_pos"]), ("joints_m", s["joints_m"]), ("vs", s["vs"]),
                     ("n1", s["n1"]), ("il1", s["il1"]), ("vd", s["vd"]), ("n2", s["n2"]), ("il2", s["il2"]), ("ndc_c", s["ndc_c"]),
                     ("centroid", s["centroid"]), ("light_R", s["light_R"]), ("light_T", s["light_T"]), ("ndc_l", s["ndc_l"]),
                     ("g_ndc_c", s["g_ndc_c"]), ("g_ndc_l", s["g_ndc_l"]), ("g_n2", s["g_n2"]), ("g_joints_m", s["g_joints_m"]), ("g_vd", s["g_vd"]),
                     ("g_light_R", s["g_light_R"]), ("g_light_T", s["g_light_T"]), ("g_v0", s["g_v0"]), ("g_joints_mm", s["g_joints_mm"]),
                     ("g_light_pos", s["g_light_pos"]), ("g_cam_T", s["g_cam_T"]), ("g_disp", self.grads["verts_disps"])):
            setattr(c, k, p(t))
        c.B, c.V0, c.E0, c.NJ, c.S = B, tp.V0, tp.E0, self.n_joints, self.S
        c.focal, c.shadow, c.has_normal_grad = self.focal, int(shadow), int(has_normal_grad)
        c.light_only = int(getattr(self, "_lean_now", False))
        return c

    def _hand_struct(self, fid, B, shadow, has_normal_grad, step=None):
        """harp_hand_front over the active lane's scratch: the one-launch front (csrc/hand_front.hip) and the three-launch back (csrc/hand_back.hip)"""
        s, p = self.s, _lib.ptr
        h = _lib.HandFront()
        if step is not None:
            h.step = step
        h.chain, h.mano, h.tables = self._chain_struct(B, shadow, has_normal_grad), self.dm.struct, self.tables
        for k, t in (("fid", fid), ("pose48", s["pose48"]), ("betas", s["betas"]), ("trans_b", s["trans_b"]), ("cam_R", s["cam_R"]),
                     ("cam_T", s["cam_T"]), ("light_pos", s["light_pos"]), ("colors", s["colors"]), ("lbs_ws", s["lbs_ws"])):
            setattr(h, k, p(t))
        h.self_shadow = int(self.self_shadow)
        return h

    def _arm_struct(self, fid, B, shadow, has_normal_grad, step=None):
        """harp_arm_front over the active lane's scratch (SMPL-X arm path): csrc/arm_front.hip"""
        s, p = self.s, _lib.ptr
        h = _lib.ArmFront()
        if step is not None:
            h.step = step
        h.chain, h.tree, h.tables = self._chain_struct(B, shadow, has_normal_grad), self.dm.struct, self.tables
        for k, t in (("fid", fid), ("pose_in", s["pose48"]), ("betas", s["betas"]), ("trans_b", s["trans_b"]), ("cam_R", s["cam_R"]),
                     ("cam_T", s["cam_T"]), ("light_pos", s["light_pos"]), ("colors", s["colors"]), ("lbs_ws", s["lbs_ws"]),
                     ("weights_T", self._weights_T)):
            setattr(h, k, p(t))
        h.self_shadow = int(self.self_shadow)
        return h

    def _mesh_forward(self, fid, B, shadow=False, front=False, step=None, stage=None):
        """frame_setup .. normals (and, fused, both projections + the light camera): fills the scratch geometry for the B frames in
        `fid` (int32 device tensor).  Returns True when the fused chain ran (projections / light camera already done).  front=True
        allows the one-launch form of the whole front (MANO path, csrc/hand_front.hip)."""
        L, s, p, st, tp = _lib.lib(), self.s, _lib.ptr, _lib.stream(), self.topo
        if front and self.fused_front and self.fused_chain:
            wide, hybrid = self.wide_front, self.hybrid_front
            if self.front_auto and not self.use_arm and stage is not None:
                # hand path, by stage (profiles/r05_wide_ab.txt item 19, fresh processes): a single-stage step (geometry only / appearance
                # only) has no long second-stream chain in front of the rasterisers, its head is on the critical path -> wide front
                # (-15 ... -25 us); the combined stage -> hybrid (hand layer wide, mesh chain one workgroup per frame: -5 us; wide: +6)
                both = bool(stage[0] and stage[1])
                wide, hybrid = (not both), both
            if self.use_arm and wide:
                self._ck(L.harp_arm_front_wide_fwd(ctypes.byref(self._arm_struct(fid, B, shadow, False, step)), p(s["chain_parts"]), st),
                         "arm_front_wide_fwd")
            elif self.use_arm:
                self._ck(L.harp_arm_front_fwd(ctypes.byref(self._arm_struct(fid, B, shadow, False, step)), st), "arm_front_fwd")
            elif hybrid and not wide:
                self._ck(L.harp_hand_front_hybrid_fwd(ctypes.byref(self._hand_struct(fid, B, shadow, False, step)), st), "hand_front_hybrid_fwd")
            elif wide:
                self._ck(L.harp_hand_front_wide_fwd(ctypes.byref(self._hand_struct(fid, B, shadow, False, step)), p(s["chain_parts"]), st),
                         "hand_front_wide_fwd")
            else:
                self._ck(L.harp_hand_front_fwd(ctypes.byref(self._hand_struct(fid, B, shadow, False, step)), st), "hand_front_fwd")
            return True
        if step is not None:
            raise RuntimeError("a folded step needs the one-launch front (fused_front)")
        self._ck(L.harp_frame_setup_fwd(ctypes.byref(self.tables), p(fid), B, self.S, self.focal, int(self.self_shadow), p(s["pose48"]),
                                        p(s["betas"]), p(s["trans_b"]), p(s["cam_R"]), p(s["cam_T"]), p(s["light_pos"]), p(s["colors"]), st),
                 "frame_setup_fwd")
        lbs_fwd = L.harp_lbs_tree_fwd if self.use_arm else L.harp_lbs_mano_fwd
        self._ck(lbs_fwd(ctypes.byref(self.dm.struct), p(s["pose48"]), p(s["betas"]), p(s["trans_b"]), B, p(s["lbs_ws"]),
                         p(s["verts_mm"]), p(s["joints_mm"]), st), "lbs_fwd")
        if self.fused_chain:
            self._ck(L.harp_mesh_chain_fwd(ctypes.byref(self._chain_struct(B, shadow, False)), st), "mesh_chain_fwd")
            return True
        self._ck(L.harp_scale(p(s["joints_mm"]), 1e-3, B * self.n_joints * 3, p(s["joints_m"]), st), "scale")          # visualize.py:46
        self._ck(L.harp_subdivide_fwd(p(s["verts_mm"]), p(tp.edges0), B, tp.V0, tp.E0, 1e-3, p(s["vs"]), st), "subdivide_fwd")
        self._ck(L.harp_vertex_normals_fwd(p(s["vs"]), p(tp.faces), p(tp.vf_off), p(tp.vf_idx), B, tp.V, p(s["n1"]), p(s["il1"]),
                                           p(self.params["verts_disps"]), p(s["vd"]), st), "normals_displace_fwd")
        self._ck(L.harp_vertex_normals_fwd(p(s["vd"]), p(tp.faces), p(tp.vf_off), p(tp.vf_idx), B, tp.V, p(s["n2"]), p(s["il2"]),
                                           None, None, st), "normals_fwd")
        return False

    @torch.no_grad()
    def compute_reference_mesh(self):
        """ARAP reference = frame-0 mesh under the initial parameters (optimize_sequence.py:429-435)."""
        fid0 = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self._mesh_forward(fid0, 1)
        if self.ref_verts is None:
            self.ref_verts = self.s["vd"][0].clone()
        else:
            self.ref_verts.copy_(self.s["vd"][0])       # same buffer: graphs captured against it stay valid

    def _shade_struct(self, B, app):
        s, tp = self.s, self.topo
        a = ops._shade_args(s["face_c"], s["ws_c"], tp, s["vd"][:B], s["n2"][:B], self.params["texture"][0], s["nmap_n"], s["light_pos"],
                            s["colors"], s["zl"] if self.self_shadow else None, s["light_R"] if self.self_shadow else None,
                            s["light_T"] if self.self_shadow else None, self.S, self.focal, (self.S / 2.0, self.S / 2.0), BG_COLOR)
        a.B = B
        # the fused photometric L1 needs no materialised image: with keep_image = False (fitting loops) the 4 MB / frame write is skipped
        a.rgb = _lib.ptr(s["rgb"]) if (self.keep_image or self.perceptual is not None) else None
        a.l1_bg_sums = _lib.ptr(self.bg_photo) if self.bg_photo is not None else None
        if self.packed_texels:
            a.texnm = _lib.ptr(self.texnm)
        for k, t in (("g_rgb", s["g_rgb"]), ("g_tex", self.grads["texture"]), ("g_nmap", s["g_nmap_n"]), ("g_verts", s["g_vd"]),
                     ("g_vnormals", s["g_n2"]), ("g_ndc", s["g_ndc_c"]), ("g_zl", s["g_zl"] if self.self_shadow else None),
                     ("g_light_pos", s["g_light_pos"]), ("g_colors", s["g_colors"]),
                     ("g_light_R", s["g_light_R"] if self.self_shadow else None), ("g_light_T", s["g_light_T"] if self.self_shadow else None)):
            setattr(a, k, _lib.ptr(t))
        if self.vert9:
            a.g_vert9 = _lib.ptr(s["g_v9"])
        if getattr(self, "_lean_now", False):            # appearance-only stage: no geometry gradients out of the shader backward
            a.g_verts = a.g_vnormals = a.g_ndc = a.g_vert9 = None
        # maps kept out of the optimiser (known_appearance, optimize_sequence.py:264-289): their gradients are not formed at all
        if "texture" in self.frozen:
            a.g_tex = None
        if "normal_map" in self.frozen:
            a.g_nmap = None
        # light-view tiles that receive a shadow-tap gradient are flagged for the depth backward (which clears what it consumes)
        a.g_zl_tiles = _lib.ptr(s["zl_tiles"]) if (self.self_shadow and self.consume_gzl and self.zl_tile_flags) else None
        if self._records_on():
            rec, cnt, cap = self._texel_record_buffers()
            a.trec, a.trec_cnt, a.trec_cap = _lib.ptr(rec), _lib.ptr(cnt), cap
            a.trec_acc_tex, a.trec_acc_nmap = _lib.ptr(self._tacc[0]), _lib.ptr(self._tacc[1])
        return a

    def _records_on(self):
        return bool(self.texel_records and not ("texture" in self.frozen and "normal_map" in self.frozen))

    def _texel_record_buffers(self):
        if self._trec is None:
            nb = _lib.lib().harp_texel_bins(self.Ht, self.Wt)
            cap = max(int(self.trec_cap_min), self.B * self.S * self.S // max(1, int(self.trec_cap_div)))
            self._trec = (torch.empty(nb * 9 * cap, dtype=torch.float32, device=self.dev),
                          torch.zeros(nb * 16 + 16, dtype=torch.int32, device=self.dev), cap)
        if self._tacc is None:                           # double accumulators of the two maps: all-zero between steps (harp_texel_finish clears what it consumes)
            self._tacc = torch.zeros(2, self.Ht * self.Wt * 3, dtype=torch.float64, device=self.dev)
        return self._trec

    def _join_maps(self):
        """the texel reduce (+ normal-map chain rule) runs on a branch of its own behind the shader backward: whoever reads the map
        gradients next (all-reduce, Adam) joins it first"""
        st, self._maps_pending = self._maps_pending, None
        if st is not None:
            torch.cuda.current_stream().wait_stream(st)

    def _can_fold(self):
        """the step's book-keeping rides in hand_front / hand_back / harp_step_prologue (`fold_step`) when those launches exist"""
        return bool(self.fold_step and self.fused_front and self.fused_chain and self.fused_back and self.overlap and self.early_terms
                    and self.schedule is not None and self._lane.get("owns_shared"))

    def forward_backward(self, coarse=True, app=True, B=None, shared_terms=True, tick=False, sched=False):
        """Enqueue forward + losses + backward for the first B (default: the lane's size) frames of the active lane; gradients land
        in self.g_buf, loss terms in loss_vec[:9] (unweighted, order LOSS_NAMES).  shared_terms=False skips everything that does not
        depend on the frames (gradient-arena zeroing, offset draw, normal-map normalisation, displacement / texture regularisers)."""
        lane = self._lane
        if getattr(self, "_shadow_state_stale", False) and not torch.cuda.is_current_stream_capturing():
            self._reset_shadow_state()
        B = lane["B"] if B is None else int(B)
        lfid, ltfid, lloss = lane["fid"], lane["tfid"], lane["loss_vec"]
        # sched: take the next row of the device schedule INSIDE this step's launches (step() passes it when _can_fold()): hand_front fetches
        # the row, the terms accumulate into loss_acc (always clean between steps), hand_back moves them to the lane's loss vector, clears
        # loss_acc and advances the schedule row and the draw counter
        fold = bool(sched)
        frame = None
        # `lean_app_stage`: in the appearance-only stage the optimiser holds texture, normal map, light position and ambient ratio
        # (optimize_sequence.py:264-310) — the reference's autograd still differentiates through the whole mesh chain and hand layer and
        # throws those gradients away.  Lean: the shader backward forms no vertex gradients, the chain backward only its light-view part
        # (-> light position), no hand-layer backward.  Same parameters after the step; g_buf's geometry segments stay zero.
        self._lean_now = bool(self.lean_app_stage and app and not coarse and shared_terms and self.fused_chain and self.perceptual is None)
        if fold:
            if B != lane["B"] or not shared_terms or not self._can_fold():
                raise RuntimeError("forward_backward(sched=True) needs the full batch, the shared terms and _can_fold()")
            frame = _lib.StepFrame()
            frame.schedule, frame.sched_row = _lib.ptr(self.schedule), _lib.ptr(self.schedule_row)
            frame.tschedule = _lib.ptr(self.tschedule) if self.tschedule is not None else None
            frame.n_rows, frame.target_offset = int(self.schedule.shape[0]), int(self.target_offset)
            frame.tfid_out, frame.clear_mesh_grads = _lib.ptr(ltfid), 1
            frame.loss, frame.loss_out, frame.n_loss = _lib.ptr(self.loss_acc), _lib.ptr(lloss), 16
            if self.accumulate_loss:
                frame.loss_w, frame.loss_total = _lib.ptr(self.w_total), _lib.ptr(self.loss_total)
            if app and self.auto_draw:
                frame.draw_counter = _lib.ptr(self.draw_counter)
            lloss = self.loss_acc
            self._loss_cleared = True
        L, s, p, ST, tp, S = _lib.lib(), self.s, _lib.ptr, _lib.stream, self.topo, self.S
        cur, side = torch.cuda.current_stream(), self._side_stream()
        V, F = tp.V, tp.F
        w = lane["w_vec"]
        wp = lambda i: w.data_ptr() + 4 * i
        lp = lambda i: lloss.data_ptr() + 4 * i
        if shared_terms and not lane.get("owns_shared"):
            self.g_buf.zero_()
            self.g_nmap_n.zero_()
        # the loss vector is the only cleared buffer the main stream touches before it joins the second one: when schedule_next has
        # cleared it already, the big slab clear moves to the second stream, off the head of the step
        fill_side = self._loss_cleared and self.early_terms and self.overlap and bool(lane.get("owns_shared"))
        self._loss_cleared = False
        # g_vd / g_joints_m: cleared by the third stream in front of the terms that accumulate into them (nothing else touches the two
        # before the streams join), otherwise together with the rest of the slab
        mesh_on_third = self.mesh_third and self.early_terms and self.overlap
        if fill_side:
            pass
        else:
            self.gs_zero.zero_()                         # main lane: one fill also covers g_buf, g_nmap_n and the loss vector
            if not mesh_on_third and not fold:
                self.gs_mesh.zero_()
            if not lane.get("owns_shared"):
                lloss.zero_()
        shadow = app and self.self_shadow
        if not self.overlap:
            side = cur                                   # single-stream mode (kernels timed with events; the lanes of a pipelined step)
        # (single-stream mode enqueues no waits at all: a stream waiting for itself is legal but has crashed hipStreamEndCapture)
        one = not self.overlap
        wait_s = lambda a, b: None if one else a.wait_stream(b)
        wait_e = lambda a, e: None if one else a.wait_event(e)
        sched_early = self.early_terms
        extra = lambda name: self._extra_stream(name) if self.overlap else cur      # further graph branches (hipGraph replays four concurrently here)
        off = self.disabled_terms
        marks = {}
        # the silhouette backward inside the camera-view raster launch (its atomics into g_ndc_c need the slab clear in front of them)
        fsb = bool(self.fused_sil_bwd and coarse and not (self.fused_bwd and app and self.perceptual is None))
        # ---- terms that depend on the parameters only (normal-map normalisation, texture regularisers, displacement regulariser) go
        #      first on the second stream: they run under the LBS / mesh chain, which is a string of small latency-bound launches
        deferred = []
        def param_terms():
            # slab clear + Adam tick + offset draw as ONE launch (harp_step_prologue); the draw counter is advanced at the end of the step by
            # hand_back (folded step) or by the launch that consumes the offsets (harp_texture_terms)
            pro = fold or (fill_side and self.fused_terms)
            draw = app and shared_terms and self.auto_draw
            if pro:
                zero = self.gs_zero[:-64] if fill_side else None
                hy, nh = (None, 0)
                if tick and (coarse or app):
                    hy, nh = (self.hyper.data_ptr(), 2) if (coarse and app) else (self.hyper.data_ptr() + (0 if coarse else 1) * self._hyper_stride, 1)
                self._ck(L.harp_step_prologue(p(zero) if zero is not None else None, zero.numel() if zero is not None else 0, hy, nh, self.seed,
                                              p(self.draw_counter), self.Ht, self.Wt, 1.0, p(self.dist_albedo) if draw else None, 2.0,
                                              p(self.dist_normal) if draw else None, ST()), "step_prologue")
                if fill_side and not mesh_on_third and not fold:
                    self.gs_mesh.zero_()
            else:
                if fill_side:
                    self.gs_zero[:-64].zero_()               # everything but the loss vector (the slab's last segment: 16 floats padded to the arena's 64-float granule — a clear that reached into the padding's front would wipe what the other streams have already added)
                    if not mesh_on_third:
                        self.gs_mesh.zero_()
                if tick:
                    self._adam_tick(coarse, app)             # only touches the hyper-parameter block: off the serial tail of the step
            if not self.consume_gzl:
                self.gs_zero_late.zero_()
            if fsb and fill_side and not one:
                marks["zero"] = torch.cuda.current_stream().record_event()      # the slab clear ran on this (the second) stream
            disp_reg = coarse and shared_terms and "vert_disp_reg" not in off
            if app and shared_terms:
                if draw and not pro:
                    self.draw_texture_offsets()
                nt = self.Ht * self.Wt
                if self.fused_terms:
                    self._ck(L.harp_normalize3_pack(p(self.params["texture"]), p(self.params["normal_map"]), nt, p(s["nmap_n"]),
                                                    p(self.texnm) if self.packed_texels else None, ST()), "normalize3_pack")
                    dr = disp_reg
                    def tex_terms():
                        self._ck(L.harp_texture_terms(p(self.params["texture"]), p(self.params["normal_map"]), p(self.uv_mask), p(self.dist_albedo),
                                                      p(self.dist_normal), self.Ht, self.Wt, 0.2, wp(7), lp(7),
                                                      None if "texture" in self.frozen else p(self.grads["texture"]), wp(8), lp(8),      # (frozen maps: loss values only)
                                                      None if "normal_map" in self.frozen else p(self.grads["normal_map"]),
                                                      p(self.params["verts_disps"]) if dr else None, V, wp(2), lp(2),
                                                      p(self.grads["verts_disps"]), p(self.draw_counter) if (draw and pro and not fold) else None, ST()),
                                 "texture_terms")
                    # `late_texture_terms`: the (atomics-bound, 40 us) regularisers are enqueued on the second stream BEHIND the light view
                    # instead of in front of it — the light view then starts at the fork, not when the regularisers are done
                    if self.late_texture_terms and self.overlap and self.early_terms and shadow:
                        deferred.append(tex_terms)
                    else:
                        tex_terms()
                    disp_reg = False
                else:
                    self._ck(L.harp_normalize3_fwd(p(self.params["normal_map"]), nt, p(s["nmap_n"]), ST()), "normalize3")
                    if self.packed_texels:
                        self._ck(L.harp_pack_texels(p(self.params["texture"]), p(s["nmap_n"]), nt, p(self.texnm), ST()), "pack_texels")
                    self._texture_terms(wp, lp)
            if disp_reg:
                self._ck(L.harp_sum_squares(p(self.params["verts_disps"]), V, wp(2), lp(2), p(self.grads["verts_disps"]), ST()), "disp_reg")

        def mesh_terms():
            kps_on, reg_on = coarse and "kps_anchor" not in off, coarse and not off.issuperset(("laplacian", "normal", "arap"))
            if kps_on and reg_on and self.fused_terms:
                self._ck(L.harp_mesh_kps_terms(p(s["vd"]), p(self.ref_verts), p(tp.nbr_off), p(tp.nbr_idx), p(tp.nc_pairs), p(tp.vp_off), p(tp.vp_idx), B, V,
                                               tp.nc_pairs.shape[0], tp.E, wp(3), lp(3), p(s["g_vd"]), p(self.init_joints), p(lfid), p(s["joints_m"]),
                                               self.n_joints, wp(1), lp(1), p(s["g_joints_m"]), ST()), "mesh_kps_terms")
                return
            if kps_on:
                self._ck(L.harp_kps_loss(p(self.init_joints), p(lfid), p(s["joints_m"]), B, self.n_joints, wp(1), lp(1), p(s["g_joints_m"]), ST()), "kps")
            if reg_on:      # (individually disabled ones carry weight 0)
                self._ck(L.harp_mesh_regularizers(p(s["vd"]), p(self.ref_verts), p(tp.nbr_off), p(tp.nbr_idx), p(tp.nc_pairs), p(tp.vp_off), p(tp.vp_idx), B, V,
                                                  tp.nc_pairs.shape[0], tp.E, wp(3), lp(3), p(s["g_vd"]), ST()), "mesh_reg")
        # Capture order matters to the replay: of the kernels that depend on one node, hipGraph keeps the FIRST-captured one on
        # that node's stream (no gap) and gives the others the next streams (~11 us of cross-stream wait each).  `graph_order` captures the
        # critical path first everywhere — hand layer before the second stream's fork, camera-view set-up before the light view's, the light
        # view before the third stream, the shader backward before the silhouette backward — so that it replays as ONE in-order stream.
        go = self.graph_order and sched_early and self.overlap and self.camera_first
        if sched_early and not go:
            wait_s(side, cur)
            with torch.cuda.stream(side):
                param_terms()
        ev0 = cur.record_event() if go else None
        fused = self._mesh_forward(lfid, B, shadow, front=True, step=frame, stage=(coarse, app))      # fused chain: both projections and the light camera are done as well
        if go:
            wait_e(side, ev0)
            with torch.cuda.stream(side):
                param_terms()
        # ---- the light-view chain (centroid -> light camera -> projection -> K=1 raster) is independent of the camera-view chain:
        #      it runs on the second HIP stream so the two rasterisations overlap (fork / join is captured into the graph); the mesh
        #      regularisers and the key-point term follow it there (the light raster is the shorter of the two)
        mesh_late = bool(not fsb and self.mesh_terms_late and sched_early and self.overlap and coarse and app and not mesh_on_third and self.perceptual is None
                         and not (self.fused_bwd and coarse and app))
        def light_view(fork=None):
            if fork is None:
                wait_s(side, cur)
            else:
                wait_e(side, fork)
            def third_branch():
                if mesh_on_third:
                    third = extra("third")
                    if fork is None:
                        wait_s(third, cur)
                    else:
                        wait_e(third, fork)
                    with torch.cuda.stream(third):
                        if not fold:                     # (a folded step: hand_front cleared its frames' slices)
                            self.gs_mesh.zero_()
                        mesh_terms()
            if not go:
                third_branch()
            with torch.cuda.stream(side):
                if sched_early and self.mesh_terms_first and not mesh_on_third and not mesh_late:
                    mesh_terms()
                if shadow:
                    if not fused:
                        self._ck(L.harp_centroid(p(s["vd"]), B, V, p(s["centroid"]), ST()), "centroid")
                        self._ck(L.harp_light_setup_fwd(p(s["centroid"]), p(s["light_pos"]), B, p(s["light_R"]), p(s["light_T"]), ST()), "light_setup")
                        self._ck(L.harp_project_fwd(p(s["vd"]), p(s["light_R"]), p(s["light_T"]), B, V, self.focal, S / 2.0, S / 2.0, S, p(s["ndc_l"]), ST()),
                                 "project_l")
                    if pair_ev is not None:
                        wait_e(torch.cuda.current_stream(), pair_ev)      # both views' set-up ran on the main stream
                    if self.keep_depth:      # the light depth map lives across steps: super-tiles that stay empty are not filled with -1 again
                        self._ck(L.harp_rasterize_fwd_keep(p(s["ndc_l"]), p(tp.faces), B, V, F, S, (0 if self.keep_image else 1) | pre, p(s["ws_l"]), p(s["face_l"]),
                                                           p(s["zl"]), p(s["zl_state"]), ST()), "raster_light")
                    else:
                        self._ck(L.harp_rasterize_fwd(p(s["ndc_l"]), p(tp.faces), B, V, F, S, (0 if self.keep_image else 2) | pre, 0.0, 1.0, p(s["ws_l"]), p(s["face_l"]),
                                                      p(s["zl"]), None, ST()),
                                 "raster_light")
                if sched_early and not self.mesh_terms_first and not mesh_on_third and not mesh_late:
                    mesh_terms()
                for fn in deferred:
                    fn()
                deferred.clear()
            if go:
                third_branch()

        def camera_view():
            # ---- camera view: projection + fused K=1 / soft-silhouette raster
            if not fused:
                self._ck(L.harp_project_fwd(p(s["vd"]), p(s["cam_R"]), p(s["cam_T"]), B, V, self.focal, S / 2.0, S / 2.0, S, p(s["ndc_c"]), ST()), "project")
            # the silhouette L1 term and its gradient are fused into the raster epilogue (no separate pass over alpha)
            # without keep_image nothing reads face ids / alpha / g_alpha in super-tiles that hold no face (shaders and the silhouette
            # backward skip them): soft = 3 leaves those 3/4 of the three images unwritten
            sparse = 0 if (self.keep_image or self.perceptual is not None) else 2
            # (geometry-only stage in the loss-only image mode: nothing reads the camera view's face ids — silhouette only)
            face_c = p(s["face_c"]) if (app or self.keep_image or not self.sil_only_raster) else None
            if fsb:
                if marks.get("zero") is not None:
                    wait_e(cur, marks["zero"])
                self._ck(L.harp_rasterize_l1_fwd_bwd(p(s["ndc_c"]), p(tp.faces), B, V, F, S, 1 | sparse | pre, ops.SIL_BLUR, ops.SIL_SIGMA, p(s["ws_c"]), face_c,
                                                     p(s["alpha"]), p(self.y_sil), p(ltfid), wp(0), lp(0), p(s["g_alpha"]), p(self.bg_sil) if sparse else None,
                                                     p(s["g_ndc_c"]), ST()), "raster_cam_fwd_bwd")
                return
            self._ck(L.harp_rasterize_l1_fwd(p(s["ndc_c"]), p(tp.faces), B, V, F, S, 1 | sparse | pre, ops.SIL_BLUR, ops.SIL_SIGMA, p(s["ws_c"]), face_c,
                                             None, p(s["alpha"]), p(self.y_sil) if coarse else None, p(ltfid), wp(0), lp(0), p(s["g_alpha"]),
                                             p(self.bg_sil) if sparse else None, ST()),
                     "raster_cam")
        # `paired_setup`: the set-up of BOTH views on the main stream, in front of the camera raster
        pair_ev, pre = None, 0
        if self.paired_setup and shadow and fused and self.camera_first:
            fork = cur.record_event()                   # fork point = end of the mesh chain (the key-point / mesh terms need no more)
            self._ck(L.harp_raster_setup_pair(p(s["ndc_c"]), ops.SIL_BLUR, p(s["ws_c"]), p(s["ndc_l"]), 0.0, p(s["ws_l"]), p(tp.faces), B, V, F, S, ST()),
                     "raster_setup_pair")
            pre = 4
            pair_ev = cur.record_event() if self.overlap else None
            camera_view()
            light_view(fork)
        elif self.camera_first:
            fork = cur.record_event()                   # fork point = end of the mesh chain, before the camera-view launches
            camera_view()
            light_view(fork)
        else:
            light_view()
            camera_view()
        wait_s(cur, side)                           # join: light depth map, regulariser gradients, normalised normal map
        if mesh_on_third:
            wait_s(cur, extra("third"))
        # both backward passes of the camera view as ONE launch (harp_shade_sil_bwd): as two kernels on two streams they cannot share a CU
        fuse_bwd = self.fused_bwd and coarse and app and self.perceptual is None
        side_used = False
        sil_after = None
        def launch_sil(ev=None):
            # the silhouette backward only needs g_alpha and the camera-view workspace: it overlaps with shading on the side stream
            if ev is None:
                wait_s(side, cur)
            else:
                wait_e(side, ev)
            with torch.cuda.stream(side):
                self._ck(L.harp_silhouette_bwd(p(tp.faces), B, V, F, S, ops.SIL_BLUR, ops.SIL_SIGMA, p(s["ws_c"]), p(s["alpha"]), p(s["g_alpha"]),
                                               p(s["g_ndc_c"]), ST()), "silhouette_bwd")
                if mesh_late:
                    mesh_terms()                # `mesh_terms_late`: beside the (latency-bound) shader backward instead of beside the (VALU-bound) rasterisers
                marks["sil"] = None if one else side.record_event()
        if coarse and not fuse_bwd and not fsb:
            if not app and self.overlap:
                # geometry-only stage: there is no shader backward to run next to — the silhouette backward stays on the critical stream
                # (two cross-stream edges, ~6 us each, off the step)
                self._ck(L.harp_silhouette_bwd(p(tp.faces), B, V, F, S, ops.SIL_BLUR, ops.SIL_SIGMA, p(s["ws_c"]), p(s["alpha"]), p(s["g_alpha"]),
                                               p(s["g_ndc_c"]), ST()), "silhouette_bwd")
            else:
                side_used = True
                if go and app:
                    sil_after = cur.record_event()       # captured right behind the shader backward (which then stays on the camera raster's stream)
                else:
                    launch_sil()
        if not sched_early:
            param_terms()
            mesh_terms()
        if app:
            a = self._shade_struct(B, app)
            if fuse_bwd or not shared_terms:
                a.trec = None                           # (the one-launch backward pair hosts the table form of the shader tile; the reduce is a shared term)
            # the photometric L1 term and its gradient are fused into the shader (no separate pass over the image)
            a.l1_target, a.l1_mask, a.l1_fid = p(self.y_true), p(self.y_sil_col), p(ltfid)
            a.l1_w, a.l1_loss, a.l1_grad = wp(6), lp(6), p(s["g_rgb"])
            # fitting loop (no image kept, no perceptual term): there is no forward shading launch — the backward pass recomputes the
            # colour anyway and forms the photometric L1 and its gradient itself (harp_shade_bwd with g_rgb == NULL)
            # (the one-launch backward pair, `fused_bwd`, instantiates the loss-only shader tile: it cannot write y_pred, so a kept
            # image goes through the forward shader there)
            fused_loss = (self.fused_loss and (not self.keep_image or (self.fused_keep and not fuse_bwd)) and self.perceptual is None
                          and self.bg_photo is not None)
            if fused_loss:
                a.g_rgb = None
            else:
                self._ck(L.harp_shade_fwd(ctypes.byref(a), ST()), "shade_fwd")
            if self.perceptual is not None:
                self._perceptual_term(B, ltfid, lloss)
        # ---- backward
        if app:
            if fuse_bwd:
                self._ck(L.harp_shade_sil_bwd(ctypes.byref(a), ops.SIL_BLUR, ops.SIL_SIGMA, p(s["alpha"]), p(s["g_alpha"]), ST()), "shade_sil_bwd")
            else:
                self._ck(L.harp_shade_bwd(ctypes.byref(a), ST()), "shade_bwd")
            if sil_after is not None:
                # `sil_late`: the silhouette backward BEHIND the shader backward (next to the depth backward) instead of next to it
                launch_sil(cur.record_event() if self.sil_late else sil_after)
            if shared_terms:
                # the normal-map chain rule (and, for N > 1, the early all-reduce of the map gradients, which overlaps with the mesh /
                # hand-layer backward) only feeds the optimiser: with `tail_side` it leaves the critical path for the second stream, which
                # is idle once the silhouette backward is done (the join in front of the mesh-chain backward already exists)
                nm_frozen = "normal_map" in self.frozen
                def maps_tail():
                    if not nm_frozen:
                        self._ck(L.harp_normalize3_bwd(p(self.params["normal_map"]), p(s["g_nmap_n"]), self.Ht * self.Wt, p(self.grads["normal_map"]), ST()),
                                 "normalize3_bwd")
                    self._allreduce_maps_early()
                # the chain rule of the normal map rides in the depth backward's launch when both exist (harp_depth_nmap_bwd)
                records = self._records_on() and not fuse_bwd
                def maps_branch():
                    rec, cnt, cap = self._texel_record_buffers()
                    at, an = (None if "texture" in self.frozen else p(self._tacc[0])), (None if nm_frozen else p(self._tacc[1]))
                    self._ck(L.harp_texel_reduce(p(rec), p(cnt), cap, self.Ht, self.Wt, at, an, B * self.S * self.S // 6, ST()), "texel_reduce")
                    # float(exact sum) -> gradient arena, the normal map's through the chain rule of its normalisation
                    self._ck(L.harp_texel_finish(at, p(self.grads["texture"]), an, p(self.grads["normal_map"]), p(self.params["normal_map"]),
                                                 self.Ht * self.Wt, ST()), "texel_finish")
                    self._allreduce_maps_early()
                nmap_in_depth = self.fused_terms and self.self_shadow and self.consume_gzl and not (self.tail_side and self.overlap) and not nm_frozen and not records
                # (with the interleaved vertex gradients the depth backward is launched below, with its riders)
                v9_riders = bool(self.vert9 and not self._lean_now and self.self_shadow and self.consume_gzl)
                ev_shade = cur.record_event() if (records and self.overlap) else None
                if records and not self.overlap:
                    maps_branch()
                elif records:
                    pass                                # (captured BEHIND the depth backward, below: the critical path keeps the shader's stream)
                elif nmap_in_depth and v9_riders:
                    pass
                elif nmap_in_depth:
                    self._ck(L.harp_depth_nmap_bwd(p(s["face_l"]), p(s["ws_l"]), p(tp.faces), p(s["g_zl"]), B, V, F, S, p(s["g_ndc_l"]),
                                                   p(self.params["normal_map"]), p(s["g_nmap_n"]), self.Ht * self.Wt, p(self.grads["normal_map"]),
                                                   p(s["zl_tiles"]) if self.zl_tile_flags else None, ST()), "depth_nmap_bwd")
                    self._allreduce_maps_early()
                elif self.tail_side and self.overlap:
                    wait_s(side, cur)
                    with torch.cuda.stream(side):
                        maps_tail()
                else:
                    maps_tail()
            else:
                nmap_in_depth, ev_shade = False, None
            v9 = bool(self.vert9 and not self._lean_now)
            depth_done = nmap_in_depth
            if v9 and self.self_shadow and self.consume_gzl:
                # the depth backward with its riders: the unpacking of the interleaved vertex gradients (and, table form, the normal map's chain rule)
                self._ck(L.harp_depth_bwd_riders(p(s["face_l"]), p(s["ws_l"]), p(tp.faces), p(s["g_zl"]), B, V, F, S, p(s["g_ndc_l"]),
                                                 p(s["zl_tiles"]) if self.zl_tile_flags else None,
                                                 p(self.params["normal_map"]) if nmap_in_depth else None, p(s["g_nmap_n"]) if nmap_in_depth else None,
                                                 self.Ht * self.Wt, p(self.grads["normal_map"]) if nmap_in_depth else None,
                                                 p(s["g_v9"]), p(s["g_vd"]), p(s["g_n2"]), p(s["g_ndc_c"]), ST()), "depth_bwd_riders")
                if nmap_in_depth:
                    self._allreduce_maps_early()
                depth_done = True
            elif v9:
                self._ck(L.harp_vert9_unpack(p(s["g_v9"]), B * V, p(s["g_vd"]), p(s["g_n2"]), p(s["g_ndc_c"]), ST()), "vert9_unpack")
            if self.self_shadow:
                if not depth_done:
                    if self.consume_gzl and self.zl_tile_flags:
                        self._ck(L.harp_depth_bwd_tiles(p(s["face_l"]), p(s["ws_l"]), p(tp.faces), p(s["g_zl"]), B, V, F, S, p(s["g_ndc_l"]), p(s["zl_tiles"]),
                                                        ST()), "depth_bwd")
                    else:
                        depth_bwd = L.harp_depth_bwd_consume if self.consume_gzl else L.harp_depth_bwd
                        self._ck(depth_bwd(p(s["face_l"]), p(s["ws_l"]), p(tp.faces), p(s["g_zl"]), B, V, F, S, p(s["g_ndc_l"]), ST()), "depth_bwd")
                if not fused:
                    self._ck(L.harp_project_bwd(p(s["vd"]), p(s["light_R"]), p(s["light_T"]), p(s["g_ndc_l"]), B, V, self.focal, S, p(s["g_vd"]),
                                                p(s["g_light_R"]), p(s["g_light_T"]), ST()), "project_bwd_l")
                    self._ck(L.harp_light_setup_bwd(p(s["centroid"]), p(s["light_pos"]), p(s["g_light_R"]), p(s["g_light_T"]), B, V, p(s["g_light_pos"]),
                                                    p(s["g_centroid"]), p(s["g_vd"]), ST()), "light_setup_bwd")
        if app and ev_shade is not None:
            # the texel reduce + normal-map chain rule follow the silhouette backward on the second stream (no further graph branch: a third
            # one made the replay run the silhouette backward BEHIND it); the mesh-chain backward then joins the silhouette backward's end
            # only, and Adam (`_join_maps`) the stream
            if side_used and marks.get("sil") is not None:
                wait_e(cur, marks["sil"])
            wait_e(side, ev_shade)
            with torch.cuda.stream(side):
                maps_branch()
            self._maps_pending = side
        elif side_used or (self.tail_side and self.overlap and app):
            wait_s(cur, side)                       # silhouette_bwd -> g_ndc_c (normal-map chain rule with tail_side)
        if fused and self.fused_front and self.fused_back:
            # the whole backward tail — mesh chain, hand layer, scatter into the parameter tables' gradient rows — as three launches
            if self.use_arm and self.wide_back:
                self._ck(L.harp_arm_back_wide_bwd(ctypes.byref(self._arm_struct(lfid, B, shadow, app, frame)), p(s["g_colors"]) if app else None,
                                                  p(s["g_pose48"]), p(s["g_betas"]), p(s["chain_parts"]), ST()), "arm_back_wide_bwd")
            elif self.use_arm:
                self._ck(L.harp_arm_back_bwd(ctypes.byref(self._arm_struct(lfid, B, shadow, app, frame)), p(s["g_colors"]) if app else None,
                                             p(s["g_pose48"]), p(s["g_betas"]), ST()), "arm_back_bwd")
            elif self.wide_back:
                self._ck(L.harp_hand_back_wide_bwd(ctypes.byref(self._hand_struct(lfid, B, shadow, app, frame)), p(s["g_colors"]) if app else None,
                                                   p(s["g_betas"]), p(s["chain_parts"]), ST()), "hand_back_wide_bwd")
            else:
                self._ck(L.harp_hand_back_bwd(ctypes.byref(self._hand_struct(lfid, B, shadow, app, frame)), p(s["g_colors"]) if app else None,
                                              p(s["g_betas"]), ST()), "hand_back_bwd")
            return
        if fold:
            raise RuntimeError("a folded step needs the fused backward tail (fused_back)")
        if fused:
            # projections, light camera, both vertex-normal passes, displacement, subdivision and the mm scaling: one launch
            self._ck(L.harp_mesh_chain_bwd(ctypes.byref(self._chain_struct(B, shadow, app)), ST()), "mesh_chain_bwd")
        else:
            self._ck(L.harp_project_bwd(p(s["vd"]), p(s["cam_R"]), p(s["cam_T"]), p(s["g_ndc_c"]), B, V, self.focal, S, p(s["g_vd"]), None,
                                        p(s["g_cam_T"]), ST()), "project_bwd_c")
            if app:
                self._ck(L.harp_vertex_normals_bwd(p(s["vd"]), p(tp.faces), p(tp.vf_off), p(tp.vf_idx), B, V, p(s["n2"]), p(s["il2"]), p(s["g_n2"]),
                                                   p(s["g_tmp"]), p(s["g_vd"]), ST()), "normals_bwd2")
            self._ck(L.harp_displace_bwd(p(s["g_vd"]), p(s["n1"]), p(self.params["verts_disps"]), B, V, p(s["g_n1"]), p(self.grads["verts_disps"]), ST()),
                     "displace_bwd")
            self._ck(L.harp_vertex_normals_bwd(p(s["vs"]), p(tp.faces), p(tp.vf_off), p(tp.vf_idx), B, V, p(s["n1"]), p(s["il1"]), p(s["g_n1"]),
                                               p(s["g_tmp"]), p(s["g_vd"]), ST()), "normals_bwd1")        # g_vs aliases g_vd (vd = vs + n d)
            self._ck(L.harp_subdivide_bwd(p(s["g_vd"]), p(tp.sub_off), p(tp.sub_idx), B, tp.V0, V, 1e-3, p(s["g_v0"]), ST()), "subdivide_bwd")
            self._ck(L.harp_scale(p(s["g_joints_m"]), 1e-3, B * self.n_joints * 3, p(s["g_joints_mm"]), ST()), "scale_bwd")
        lbs_bwd = L.harp_lbs_tree_bwd if self.use_arm else L.harp_lbs_mano_bwd
        if not self._lean_now:                          # (lean: g_pose48 / g_betas / g_trans_b stay the zeros of the slab clear)
            self._ck(lbs_bwd(ctypes.byref(self.dm.struct), p(s["pose48"]), p(s["betas"]), p(s["trans_b"]), B, p(s["lbs_ws"]),
                             p(s["g_v0"]), p(s["g_joints_mm"]), p(s["g_pose48"]), p(s["g_betas"]), p(s["g_trans_b"]), ST()), "lbs_bwd")
        self._ck(L.harp_frame_setup_bwd(ctypes.byref(self.tables), p(lfid), B, S, self.focal, int(self.self_shadow), p(s["g_pose48"]),
                                        p(s["g_betas"]), p(s["g_trans_b"]), p(s["g_cam_T"]), p(s["g_light_pos"]) if app else None,
                                        p(s["g_colors"]) if app else None, ST()), "frame_setup_bwd")

    # ---- optional perceptual term (optimize_sequence.py:405, 546-547) -------------------------------
    def set_perceptual(self, vgg, weight=1.0, cache_bytes=128 << 30, precision=0, bounded=True):
        """Add `weight * L1(vgg(y_pred * mask), vgg(y_true * mask))` to the appearance stage.  `vgg`: harp_amd.model.vgg.Vgg16Features
        (the filters; None removes the term).  The ten convolutions, their data gradients and everything between them run on the HIP
        kernels of csrc/conv.hip (harp_vgg16_term: 21 launches, captured into the step's hipGraph like every other launch).  precision:
        0 = float32 MFMA (a float32 fma chain, what the parity tests anchor on), 1 = three-term bf16 split with float32 accumulation
        (~16 mantissa bits per product; the reference's own stack runs these convolutions in TF32).
        The target frames' features do not change during a fit; what is kept in HBM depends on `cache_bytes`:
          * ALL 13 activation maps of every resident frame (300 floats per pixel, 307 MB per 512x512 frame — 256 frames are 79 GB of the
            288 GB) -> `bounded` mode: y_pred * mask and y_true * mask are identical outside the mask's support, so the stack runs only in
            the tiles (16 pixels a side, 8 at S/4 and S/8) the support reaches through the receptive field and reads the cached target activations next to them
            (same loss and gradient; csrc/conv.hip, model/vgg_hip.active_tiles);
          * else the four tap maps (126 MB per frame): one full forward + backward over the B rendered images per step;
          * else nothing: every step also recomputes the target features of its B frames."""
        from .model.vgg_hip import Vgg16Hip, activation_shapes, active_tiles, tap_shapes
        self._vgg_module = vgg
        self._vgg_cache_bytes = int(cache_bytes)         # the caller's budget (set_targets re-invokes with it)
        self._vgg_precision = int(precision)
        self._vgg_bounded = bool(bounded)
        self.perceptual = None if vgg is None else Vgg16Hip(vgg, self.dev, precision)
        self.graph_perceptual = True
        self.perceptual_weight = float(weight)
        self._vgg_cache = None
        self._vgg_bound = None
        self._vgg_step_feats = None
        self._graphs = {}
        if vgg is None or self.y_true is None:
            return
        T, S = self.y_true.shape[0], self.S
        full = 4 * sum(h * w * c for h, w, c in activation_shapes(S))
        taps = 4 * sum(h * w * c for h, w, c in tap_shapes(S))
        # the budget is the caller's figure, but never more than 80 % of what the device has free right now (a smaller device, or several
        # ranks sharing one: each would otherwise take the full default); a tier whose allocation fails anyway falls through to the next
        if self._records_on():
            self._texel_record_buffers()                 # (allocated before the cache is sized against what the device has free)
        torch.cuda.synchronize(self.dev)
        budget = min(int(cache_bytes), int(0.8 * torch.cuda.mem_get_info(self.dev)[0]))
        tiers = []
        if bounded and T * full <= budget:
            tiers.append((activation_shapes(S), True))
        if T * taps <= budget:
            tiers.append((tap_shapes(S), False))
        for shapes, all_slots in tiers:
            try:
                self._vgg_cache = [torch.empty((T,) + shp, device=self.dev) for shp in shapes]
            except torch.OutOfMemoryError:
                self._vgg_cache = None
                torch.cuda.empty_cache()
                continue
            if all_slots:
                self._vgg_bound = active_tiles(self.y_sil_col)
            break
        if self._vgg_cache is None:
            self._vgg_step_feats = [torch.empty((self.B,) + shp, device=self.dev) for shp in tap_shapes(S)]
            return
        for t0 in range(0, T, self.B):
            rows = torch.arange(t0, min(T, t0 + self.B), device=self.dev, dtype=torch.int32)
            self.perceptual.features(self.y_true, self.y_sil_col, rows, out=[c[t0:t0 + rows.shape[0]] for c in self._vgg_cache], all_slots=all_slots)

    def _perceptual_term(self, B, ltfid, lloss):
        """the term's value into slot 9 of the loss vector; its gradient joins the photometric gradient the shader backward consumes (that
        buffer is only defined at covered pixels — the fused L1 writes nothing elsewhere — hence the `covered` argument)"""
        s = self.s
        rows = ltfid[:B]
        if self._vgg_cache is not None:
            target, by_row = self._vgg_cache, 1
        else:
            target, by_row = [f[:B] for f in self._vgg_step_feats], 0
            self.perceptual.features(self.y_true, self.y_sil_col, rows, out=target)
        # (the batch in parts on as many streams: the term's 21 dependent launches fill each other's last rounds of workgroups)
        more = [self._extra_stream("vgg%d" % i) for i in range(min(self.vgg_streams, 4, B) - 1)] if (self.overlap and by_row) else []
        self.perceptual.term(s["rgb"][:B], self.y_true, self.y_sil_col, rows, target, by_row, s["g_rgb"][:B], lloss[9:10], weight=self.perceptual_weight,
                             covered=s["face_c"][:B], bound=self._vgg_bound, side_streams=more)

    def _extra_stream(self, name):
        lane = self._lane
        if lane.get(name) is None:
            lane[name] = torch.cuda.Stream(device=self.dev)
        return lane[name]


    def _side_stream(self):
        lane = self._lane
        if lane["side"] is None:
            lane["side"] = torch.cuda.Stream(device=self.dev)
        return lane["side"]

    def _texture_terms(self, wp, lp):
        """albedo_reg + normal_reg (loss/texture_reg.py) with their gradients, straight into the gradient arena"""
        L, p, st = _lib.lib(), _lib.ptr, _lib.stream()
        self._ck(L.harp_texture_smooth_reg(p(self.params["texture"]), p(self.dist_albedo), p(self.uv_mask), self.Ht, self.Wt, wp(7), lp(7),
                                           p(self.grads["texture"]), st), "albedo_reg")
        self._ck(L.harp_close_to_z_reg(p(self.params["normal_map"]), self.Ht, self.Wt, 0.2, wp(8), lp(8), p(self.grads["normal_map"]), st), "close_z")
        self._ck(L.harp_texture_smooth_reg(p(self.params["normal_map"]), p(self.dist_normal), p(self.uv_mask), self.Ht, self.Wt, wp(8), lp(8),
                                           p(self.grads["normal_map"]), st), "normal_smooth")

    def _dist_on(self):
        return self.world > 1 or self.force_allreduce

    def set_comm(self, comm):
        """harp_amd.dist.RcclComm (or None): the gradient bucket is then reduced by `harp_allreduce_flat` on the step's own streams — a
        plain enqueue, captured into the step's hipGraph like every kernel — instead of through torch.distributed."""
        self.comm = comm
        self._graphs = {}
        if comm is not None:
            comm._engines.add(self)                      # RcclComm.destroy() un-sets itself here (captured graphs hold its raw handle)

    def _comm_stream(self):
        if getattr(self, "_cstream", None) is None:
            self._cstream = torch.cuda.Stream(device=self.dev)
        return self._cstream

    def _allreduce_maps_early(self):
        """The texture + normal-map gradients (6.29 of the 6.36 MB bucket) are final once the shading backward and normalize3_bwd
        are enqueued, ~0.25 ms before the mesh / LBS backward tail ends: their all-reduce is started there and overlaps with that
        tail; `allreduce()` then only has the small remainder [pose .. amb_ratio] left to send.  RcclComm: both collectives go to one
        communication stream (forked from / joined into the step's stream with events — all capturable), so the two operations on the
        communicator stay ordered.  torch.distributed: async_op on the process group's own stream."""
        if not self._dist_on() or not self.overlap_allreduce:
            return
        o = self.arena.offsets["texture"][0]
        e = self.arena.span("texture", "normal_map")
        if self.comm is not None:
            cur, cs = torch.cuda.current_stream(), self._comm_stream()
            cs.wait_stream(cur)
            self.comm.allreduce(self.g_buf[o:o + e[1]], stream=cs.cuda_stream)
            self._early_from = o
            self._early_work = "rccl"
            return
        if torch.cuda.is_current_stream_capturing():
            return
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return
        self._early_work = dist.all_reduce(self.g_buf[o:o + e[1]], async_op=True)
        self._early_from = o

    def allreduce(self):
        if not self._dist_on():
            return                                       # (the maps' branch stays open: adam() joins it — behind the maps' own update, `split_adam`)
        self._join_maps()
        from .dist import allreduce_flat
        o, n = self.opt_span
        work, self._early_work = self._early_work, None
        if self.comm is not None:
            if work is not None:
                cur, cs = torch.cuda.current_stream(), self._comm_stream()
                cs.wait_stream(cur)                                 # the remainder is final only now (end of the backward tail)
                self.comm.allreduce(self.g_buf[o:self._early_from], stream=cs.cuda_stream)
                cur.wait_stream(cs)
            else:
                self.comm.allreduce(self.g_buf[o:o + n])            # one flat bucket (sum); 1/world is applied in the Adam kernel
            return
        if work is not None:
            allreduce_flat(self.g_buf[o:self._early_from])          # everything before the maps (they are the tail of the bucket)
            work.wait()                                             # current stream waits for the early collective
        else:
            allreduce_flat(self.g_buf[o:o + n])

    def _adam_tick(self, coarse, app):
        """advance step / bias corrections of the stage's optimiser(s) — any time before `adam(..., tick=False)` of the same step"""
        L, st = _lib.lib(), _lib.stream()
        if coarse and app:                               # the two hyper-parameter structs are adjacent: one launch for both
            self._ck(L.harp_adam_tick(self.hyper.data_ptr(), 2, st), "adam_tick")
        elif coarse or app:
            self._ck(L.harp_adam_tick(self.hyper.data_ptr() + (0 if coarse else 1) * self._hyper_stride, 1, st), "adam_tick")

    def adam(self, coarse=True, app=True, tick=True):
        # the maps are 99.99 % of the optimised elements and their gradients are final on the second stream (texel reduce -> finish) well before
        # the backward tail ends: their Adam update runs THERE, and the launch that ends the step only carries the ~15 k other parameters
        # (8 -> 4 us at the very end of the critical path).  Single rank, nothing frozen, hyper-parameters already ticked by the prologue.
        maps = self._maps_pending
        if (self.split_adam and maps is not None and app and not tick and not self.frozen and not self._dist_on()):
            L, h1 = _lib.lib(), self.hyper.data_ptr() + self._hyper_stride
            bufs = (self.p_buf.data_ptr(), self.g_buf.data_ptr(), self.m_buf.data_ptr(), self.v_buf.data_ptr())
            om, nm = self.arena.span("texture", "normal_map")
            with torch.cuda.stream(maps):
                self._ck(L.harp_adam_apply(*(b + 4 * om for b in bufs), nm, h1, _lib.stream()), "adam_apply(maps)")
            self._join_maps()
            osm, nsm = self.arena.span("light_positions", "amb_ratio")
            if coarse:
                (o0, n0) = self.coarse_span
                self._ck(L.harp_adam_apply2(*bufs, o0, n0, osm, nsm, self.hyper.data_ptr(), _lib.stream()), "adam_apply2(small)")
            else:
                self._ck(L.harp_adam_apply(*(b + 4 * osm for b in bufs), nsm, h1, _lib.stream()), "adam_apply(small)")
            return
        self._join_maps()
        L, p, st = _lib.lib(), _lib.ptr, _lib.stream()
        # parameters outside the reference's optimiser groups (known_appearance: shape / displacement / texture / normal map,
        # optimize_sequence.py:264-289) keep a zero gradient: with m = v = 0 the dense Adam update of such an element is exactly 0
        for k in self.frozen:
            self.grads[k].zero_()
        if tick:
            self._adam_tick(coarse, app)
        bufs = (self.p_buf.data_ptr(), self.g_buf.data_ptr(), self.m_buf.data_ptr(), self.v_buf.data_ptr())
        if coarse and app:                               # both groups in one launch
            (o0, n0), (o1, n1) = self.coarse_span, self.app_span
            self._ck(L.harp_adam_apply2(*bufs, o0, n0, o1, n1, self.hyper.data_ptr(), st), "adam_apply2")
        elif coarse or app:
            o, n = self.coarse_span if coarse else self.app_span
            h = self.hyper.data_ptr() + (0 if coarse else 1) * self._hyper_stride
            self._ck(L.harp_adam_apply(*(b + 4 * o for b in bufs), n, h, st), "adam_apply")

    # ------------------------------------------------------------------------------------------------
    def set_stage(self, coarse, app):
        w = torch.zeros(16)
        for i, k in enumerate(LOSS_NAMES):
            if ((coarse and k in COARSE_TERMS) or (app and k in APP_TERMS)) and k not in self.disabled_terms:
                w[i] = LOSS_WEIGHTS[k]
        self.w_vec.copy_(w.to(self.dev))
        # the weights of the step's sum_loss (optimize_sequence.py:553-559): the same, plus the perceptual term's in slot 9
        if app and self.perceptual is not None:
            w[9] = self.perceptual_weight
        self.w_total.copy_(w.to(self.dev))

    def set_disabled_terms(self, names):
        """Leave loss terms out of the objective: weight 0, kernels not launched, loss value reported as 0.  The reference fits a test
        sequence with a known appearance (`known_appearance`) WITHOUT the key-point anchor and the mesh regularisers
        (optimize_sequence.py:523, 531: kps_anchor, vert_disp_reg, laplacian, normal, arap)."""
        names = frozenset(names)
        unknown = names - set(LOSS_NAMES)
        if unknown:
            raise ValueError(f"unknown loss terms {sorted(unknown)}")
        self.disabled_terms = names
        self._stage = None                               # weights are re-uploaded by the next step()
        self._graphs = {}

    def set_lr(self, lr_coarse=None, lr_app=None):
        """host -> device hyper block (ReduceLROnPlateau lives on the host, optimize_sequence.py:309, 581-582)"""
        new = (None if lr_coarse is None else float(lr_coarse), None if lr_app is None else float(lr_app))
        last = getattr(self, "_lr_set", (None, None))
        if all(n is None or n == l for n, l in zip(new, last)):
            return                                       # unchanged since the last call (every epoch without a plateau): no D2H + H2D round trip
        h = self.hyper.cpu().numpy().view(self.hyper_np.dtype)
        if lr_coarse is not None:
            h["lr"][0] = lr_coarse
        if lr_app is not None:
            h["lr"][1] = lr_app
        self.hyper.copy_(torch.from_numpy(h.view(np.uint8)).to(self.dev))
        self._lr_set = tuple(n if n is not None else l for n, l in zip(new, last))

    def draw_texture_offsets(self):
        """the random neighbour offsets of albedo_reg (std 1) / smooth_texture_reg (std 2), loss/texture_reg.py:15, 51 — drawn on the
        device by a counter-based generator seeded identically on every rank (graph-replayable: the counter is device memory)."""
        L, p, st = _lib.lib(), _lib.ptr, _lib.stream()
        self._ck(L.harp_draw_texture_offsets(self.seed, p(self.draw_counter), self.Ht, self.Wt, 1.0, p(self.dist_albedo), 2.0, p(self.dist_normal), st),
                 "draw_offsets")

    def set_schedule(self, schedule, tschedule=None):
        """(n_rows, batch_size) global frame ids, kept on the device: `step(None, ...)` then takes the next row (wrapping around)
        inside the step's hipGraph, so a replay needs no host-side copy at all (the reference's DataLoader hands a host tensor over
        every step, optimize_sequence.py:399, :446).  tschedule: the rows of the resident targets these frames compare against, same
        shape (default fid - target_offset, i.e. targets stored in frame order) — for a dataset that holds a subset / another order of
        the frames.  A schedule of the SAME shape as the current one is written into the buffers the captured step graphs already read
        (a new epoch's shuffle costs two small copies, no re-capture); the row counter starts at 0 again."""
        sch = torch.as_tensor(schedule).to(torch.int32).to(self.dev).contiguous()
        if sch.dim() != 2 or sch.shape[1] != self.B:
            raise ValueError(f"schedule must be (n_rows, {self.B}), got {tuple(sch.shape)}")
        tsch = None
        if tschedule is not None:
            tsch = torch.as_tensor(tschedule).to(torch.int32).to(self.dev).contiguous()
            if tsch.shape != sch.shape:
                raise ValueError(f"tschedule must have the schedule's shape {tuple(sch.shape)}, got {tuple(tsch.shape)}")
        same = (self.schedule is not None and self.schedule.shape == sch.shape and (self.tschedule is None) == (tsch is None))
        if same:
            self.schedule.copy_(sch)
            if tsch is not None:
                self.tschedule.copy_(tsch)
            self.schedule_row.zero_()
            return
        self.schedule, self.tschedule = sch, tsch
        self.schedule_row = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self._graphs = {}                               # graphs captured against an older schedule buffer are stale

    def _schedule_next(self):
        # the same launch clears the loss vector: with it gone from the slab clear, that one runs on the second stream (forward_backward)
        self._ck(_lib.lib().harp_schedule_next_rows(_lib.ptr(self.schedule), _lib.ptr(self.tschedule) if self.tschedule is not None else None,
                                                    int(self.schedule.shape[0]), self.B, self.target_offset, _lib.ptr(self.schedule_row),
                                                    _lib.ptr(self.fid), _lib.ptr(self.tfid), _lib.ptr(self.loss_vec), 16, _lib.stream()),
                 "schedule_next")
        self._loss_cleared = True

    def step(self, fid, coarse=True, app=True, use_graph=True, tfid=None):
        """One optimisation step on the frames `fid` (global frame ids = rows of the parameter tables, length <= batch_size; a shorter —
        last, partial — batch, optimize_sequence.py:396-399, replays a graph captured for its size).  fid=None: the next row of the schedule given to
        `set_schedule`.  tfid: rows of the resident targets these frames compare against (default fid - target_offset, i.e. targets
        stored in frame order); a dataset that holds a subset / another order of the frames passes its own item indices."""
        scheduled = fid is None
        if scheduled:
            if self.schedule is None:
                raise ValueError("step(None, ...) needs set_schedule() first")
            n = self.B
        else:
            fid = torch.as_tensor(fid)
            n = int(fid.shape[0])
        if n > self.B:
            raise ValueError(f"batch of {n} frames exceeds the engine's batch_size {self.B}")
        if scheduled:
            pass
        else:
            t = (fid - self.target_offset) if tfid is None else torch.as_tensor(tfid)
            if int(t.shape[0]) != n:
                raise ValueError("tfid must have one entry per frame of the batch")
            # (a device-resident batch costs no host sync at all)
            self.fid[:n].copy_(fid.to(torch.int32).to(self.dev), non_blocking=True)
            self.tfid[:n].copy_(t.to(torch.int32).to(self.dev), non_blocking=True)
        # a flipped consume_gzl / keep_depth left g_zl / zl_state in the other mode's state: a cached graph replays without passing through
        # forward_backward, so the invariant is re-established here, in front of the graph lookup
        if getattr(self, "_shadow_state_stale", False):
            self._reset_shadow_state()
        key = (coarse, app)
        if self._stage != key:
            self.set_stage(coarse, app)
            self._stage = key
        fold = scheduled and self._can_fold()
        fb0 = lambda: self.forward_backward(coarse, app, B=n, tick=True, sched=fold)
        fb = (lambda: (self._schedule_next(), fb0())) if (scheduled and not fold) else fb0
        dist_on = self._dist_on()
        graph_ok = (not dist_on) or self.comm is not None or self.graph_collectives
        # (a shorter — last, partial — batch of an epoch gets a graph of its own: the batch size is part of the key)
        if not use_graph or not graph_ok or (app and self.perceptual is not None and not self.graph_perceptual):
            fb()
            self.allreduce()
            self.adam(coarse, app, tick=False)
            if self.accumulate_loss and not fold:
                self.loss_total.add_(torch.dot(self.loss_vec, self.w_total))
            return
        # every switch the enqueued launch sequence depends on is part of the key: flipping one re-captures instead of replaying a
        # graph recorded for another configuration
        gkey = (coarse, app, scheduled, n, self.keep_image, self.fused_loss, self.self_shadow, tuple(self.frozen), self.overlap, self.early_terms,
                self.packed_texels, self.fused_keep, self.mesh_third, self.graph_order, fold, self.fused_terms, self.zl_tile_flags, self.accumulate_loss, self.lean_app_stage, self.sil_only_raster, self.auto_draw, self.mesh_terms_late, self.sil_late, self.paired_setup, self.late_texture_terms, self.mesh_terms_first, self.camera_first, self.tail_side, self.fused_bwd, self.fused_chain, self.fused_front, self.wide_front, self.hybrid_front, self.front_auto, self.wide_back, self.fused_back, self.consume_gzl, self.keep_depth, self.texel_records, self.fused_sil_bwd, self.split_adam, self.vert9, self.vgg_streams, dist_on, self.overlap_allreduce,
                self.comm is not None, self.perceptual is not None and app)
        g = self._graphs.get(gkey)
        if g is None:
            # warm-up on a side stream, then capture (torch's documented recipe)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            row = self.schedule_row.clone() if scheduled else None
            hyper, draws, ltot = self.hyper.clone(), self.draw_counter.clone(), self.loss_total.clone()
            with torch.cuda.stream(side):
                fb()
                self.allreduce()                         # completes (and clears) the early all-reduce the warm-up pass started
            torch.cuda.current_stream().wait_stream(side)
            self.hyper.copy_(hyper)                      # the warm-up pass must not advance the optimiser's step count ...
            self.draw_counter.copy_(draws)               # ... nor the texture-offset generator (same draws as an eager run with this seed)
            self.loss_total.copy_(ltot)                  # ... nor count its losses into the epoch's sum (accumulate_loss)
            if scheduled:
                self.schedule_row.copy_(row)             # ... nor consume a schedule row
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            # N > 1: other threads of the process (RCCL's proxy, torch's process-group watchdog) make HIP calls of their own while this
            # thread captures; only this thread's calls belong to the capture
            with torch.cuda.graph(g, capture_error_mode="thread_local" if dist_on else "global"):
                fb()
                self.allreduce()                         # no-op for a single rank; RCCL all-reduce is captured into the graph otherwise
                self.adam(coarse, app, tick=False)
            self._graphs[gkey] = g
            # the capture itself does not execute; fall through to the first replay
        g.replay()
        if self.accumulate_loss and not fold:
            self.loss_total.add_(torch.dot(self.loss_vec, self.w_total))

    def losses(self):
        """dict of the last step's unweighted loss terms (one D2H copy; call sparingly)."""
        v = self.loss_vec.cpu().tolist()
        out = {k: v[i] for i, k in enumerate(LOSS_NAMES)}
        if self.perceptual is not None:
            out["vgg"] = v[9]
        return out
