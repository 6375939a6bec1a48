"""ctypes binding of the C-ABI library (include/harp_hip.h).  No CPU fallback: if the library is missing
or a call fails, raise — the product path must never silently run anything else."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HARP_LIB_PATH") or os.path.join(_HERE, "csrc", "libharp_hip.so")   # override: A/B of kernel variants

_vp, _i, _f, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t

# name -> (restype, argtypes); must list every symbol include/harp_hip.h declares (tests/test_abi.py checks)
SIGNATURES = {
    "harp_rasterize_ws_bytes": (_sz, [_i, _i, _i]),
    "harp_rasterize_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "harp_silhouette_bwd": (_i, [_vp, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp]),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"HIP library not built: {LIB_PATH} (run `python -m harp_amd.build`); "
                               "harp_amd has no CPU fallback")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def ptr(t):
    """Raw device pointer of a contiguous CUDA(HIP) tensor, or NULL for None."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("harp_amd ops need HIP device tensors (no CPU path)")
    if not t.is_contiguous():
        raise RuntimeError("harp_amd ops need contiguous tensors")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def check(code, what):
    if code != 0:
        raise RuntimeError(f"{what} failed with status {code}")


class ShadeArgs(ctypes.Structure):
    """mirror of `harp_shade_args` (include/harp_hip.h)"""
    _fields_ = ([(n, _vp) for n in ("face_id", "recs", "faces", "faces_uvs", "verts_uvs", "verts", "vnormals", "tex", "nmap",
                                    "light_pos", "colors", "zl", "light_R", "light_T")] +
                [(n, _i) for n in ("B", "V", "F", "S", "Ht", "Wt")] +
                [(n, _f) for n in ("focal", "ppx", "ppy")] + [("bg", _f * 3)] +
                [(n, _vp) for n in ("rgb", "g_rgb", "g_tex", "g_nmap", "g_verts", "g_vnormals", "g_ndc", "g_zl", "g_light_pos",
                                    "g_colors", "g_light_R", "g_light_T")] + [("debug_skip", _i)] +
                [(n, _vp) for n in ("l1_target", "l1_mask", "l1_fid", "l1_w", "l1_loss", "l1_grad")] + [("l1_inv", _f), ("texnm", _vp), ("l1_bg_sums", _vp), ("g_zl_tiles", _vp),
                 ("trec", _vp), ("trec_cnt", _vp), ("trec_cap", _i), ("trec_acc_tex", _vp), ("trec_acc_nmap", _vp), ("g_vert9", _vp)])


SIGNATURES.update({
    "harp_depth_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "harp_depth_bwd_consume": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "harp_shade_fwd": (_i, [ctypes.POINTER(ShadeArgs), _vp]),
    "harp_shade_bwd": (_i, [ctypes.POINTER(ShadeArgs), _vp]),
    "harp_texel_bins": (_i, [_i, _i]),
    "harp_texel_reduce": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp]),
    "harp_texel_finish": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "harp_pack_texels": (_i, [_vp, _vp, _i, _vp, _vp]),
    "harp_normalize3_pack": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "harp_depth_nmap_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "harp_depth_bwd_tiles": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "harp_depth_bwd_riders": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "harp_vert9_unpack": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "harp_subdivide_fwd": (_i, [_vp, _vp, _i, _i, _i, _f, _vp, _vp]),
    "harp_subdivide_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _vp, _vp]),
    "harp_vertex_normals_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "harp_vertex_normals_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "harp_displace_bwd": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "harp_project_fwd": (_i, [_vp, _vp, _vp, _i, _i, _f, _f, _f, _i, _vp, _vp]),
    "harp_project_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _i, _vp, _vp, _vp, _vp]),
    "harp_centroid": (_i, [_vp, _i, _i, _vp, _vp]),
})


class ManoModel(ctypes.Structure):
    """mirror of `harp_mano_model` (include/harp_hip.h)"""
    _fields_ = [(n, _vp) for n in ("v_template", "shapedirs_T", "posedirs_T", "posedirs", "J_template", "J_dirs", "weights", "hands_mean")]


_mp = ctypes.POINTER(ManoModel)
SIGNATURES.update({
    "harp_lbs_mano_ws_floats": (_sz, [_i]),
    "harp_lbs_mano_fwd": (_i, [_mp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "harp_lbs_mano_bwd": (_i, [_mp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "harp_image_l1": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "harp_kps_loss": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "harp_mesh_regularizers": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "harp_sum_squares": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "harp_mse": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "harp_texture_smooth_reg": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "harp_mesh_kps_terms": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "harp_texture_terms": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "harp_close_to_z_reg": (_i, [_vp, _i, _i, _f, _vp, _vp, _vp, _vp]),
    "harp_normalize3_fwd": (_i, [_vp, _i, _vp, _vp]),
    "harp_normalize3_bwd": (_i, [_vp, _vp, _i, _vp, _vp]),
    "harp_adam_step": (_i, [_vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _i, _f, _vp]),
})


class FrameTables(ctypes.Structure):
    """mirror of `harp_frame_tables` (include/harp_hip.h)"""
    _fields_ = ([(n, _vp) for n in ("pose", "rot", "trans", "cam", "shape", "light_positions", "amb_ratio", "g_pose", "g_rot", "g_trans",
                                    "g_cam", "g_shape", "g_light_positions", "g_amb_ratio")] + [("share_light", _i)] +
                [("wrist_pose", _vp), ("g_wrist_pose", _vp), ("n_betas_out", _i)])


_tp = ctypes.POINTER(FrameTables)
SIGNATURES.update({
    "harp_frame_setup_fwd": (_i, [_tp, _vp, _i, _i, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "harp_frame_setup_bwd": (_i, [_tp, _vp, _i, _i, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "harp_light_setup_fwd": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "harp_light_setup_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "harp_scale": (_i, [_vp, _f, _i, _vp, _vp]),
    "harp_schedule_next": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "harp_schedule_next_rows": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp]),
})

SIGNATURES.update({
    "harp_adam_tick": (_i, [_vp, _i, _vp]),
    "harp_step_prologue": (_i, [_vp, _sz, _vp, _i, ctypes.c_uint, _vp, _i, _i, _f, _vp, _f, _vp, _vp]),
    "harp_adam_apply": (_i, [_vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "harp_adam_apply2": (_i, [_vp, _vp, _vp, _vp, _sz, _sz, _sz, _sz, _vp, _vp]),
})


class TreeModel(ctypes.Structure):
    """mirror of `harp_tree_model` (include/harp_hip.h)"""
    _fields_ = ([(n, _i) for n in ("NV", "NJ", "NB")] +
                [(n, _vp) for n in ("v_template", "shapedirs_T", "posedirs_T", "posedirs", "J_template", "J_dirs", "weights", "pose_mean", "parents",
                                    "pose_src")] + [("n_pose_in", _i), ("center_joint", _i), ("n_joints_out", _i), ("joint_src", _vp)])


_trp = ctypes.POINTER(TreeModel)
SIGNATURES.update({
    "harp_lbs_tree_ws_floats": (_sz, [_trp, _i]),
    "harp_lbs_tree_fwd": (_i, [_trp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "harp_lbs_tree_bwd": (_i, [_trp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
})


class MeshChain(ctypes.Structure):
    """mirror of `harp_mesh_chain` (include/harp_hip.h)"""
    _fields_ = ([(n, _vp) for n in ("edges0", "vf_off", "vf_tri", "sub_off", "sub_idx", "disp")] +
                [(n, _i) for n in ("B", "V0", "E0", "NJ", "S")] + [("focal", _f), ("shadow", _i), ("has_normal_grad", _i), ("light_only", _i)] +
                [(n, _vp) for n in ("verts_mm", "joints_mm", "cam_R", "cam_T", "light_pos",
                                    "joints_m", "vs", "n1", "il1", "vd", "n2", "il2", "ndc_c", "centroid", "light_R", "light_T", "ndc_l",
                                    "g_ndc_c", "g_ndc_l", "g_n2", "g_joints_m", "g_vd", "g_light_R", "g_light_T",
                                    "g_v0", "g_joints_mm", "g_light_pos", "g_cam_T", "g_disp")])


SIGNATURES["harp_mesh_chain_max_vertices"] = (_i, [])
SIGNATURES["harp_mesh_chain_fwd"] = (_i, [ctypes.POINTER(MeshChain), _vp])
SIGNATURES["harp_mesh_chain_bwd"] = (_i, [ctypes.POINTER(MeshChain), _vp])
SIGNATURES["harp_mesh_chain_wide_ws_floats"] = (_sz, [_i, _i])
SIGNATURES["harp_mesh_chain_bwd_wide"] = (_i, [ctypes.POINTER(MeshChain), _vp, _vp])
SIGNATURES["harp_mesh_chain_fwd_wide"] = (_i, [ctypes.POINTER(MeshChain), _i, _vp, _vp])
SIGNATURES["harp_draw_texture_offsets"] = (_i, [ctypes.c_uint, _vp, _i, _i, _f, _vp, _f, _vp, _vp])
SIGNATURES["harp_raster_setup_pair"] = (_i, [_vp, _f, _vp, _vp, _f, _vp, _vp, _i, _i, _i, _i, _vp])
SIGNATURES["harp_rasterize_fwd_keep"] = (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp])
SIGNATURES["harp_rasterize_l1_fwd"] = (_i, [_vp, _vp, _i, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp])
SIGNATURES["harp_rasterize_l1_fwd_bwd"] = (_i, [_vp, _vp, _i, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp])

# data-parallel exchange (csrc/comm.hip): RCCL bound at run time, all-reduce enqueued on the caller's stream
SIGNATURES.update({
    "harp_comm_unique_id": (_i, [_vp]),
    "harp_comm_create": (_i, [_vp, _i, _i, ctypes.POINTER(_vp)]),
    "harp_comm_destroy": (_i, [_vp]),
    "harp_allreduce_flat": (_i, [_vp, _vp, _sz, _vp]),
})
COMM_ID_BYTES = 128
SIGNATURES.update({
    "harp_rasterize_fragments_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "harp_rasterize_fragments_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _vp]),
})
SIGNATURES["harp_shade_sil_bwd"] = (_i, [ctypes.POINTER(ShadeArgs), _f, _f, _vp, _vp, _vp])


class StepFrame(ctypes.Structure):
    """mirror of `harp_step_frame` (include/harp_hip.h)"""
    _fields_ = [("schedule", _vp), ("tschedule", _vp), ("sched_row", _vp), ("n_rows", _i), ("target_offset", _i), ("tfid_out", _vp), ("clear_mesh_grads", _i),
                ("loss", _vp), ("loss_out", _vp), ("n_loss", _i), ("draw_counter", _vp), ("loss_w", _vp), ("loss_total", _vp)]


class HandFront(ctypes.Structure):
    """mirror of `harp_hand_front` (include/harp_hip.h)"""
    _fields_ = ([("chain", MeshChain), ("mano", ManoModel), ("tables", FrameTables), ("fid", _vp)] +
                [(n, _vp) for n in ("pose48", "betas", "trans_b", "cam_R", "cam_T", "light_pos", "colors", "lbs_ws")] + [("self_shadow", _i),
                 ("step", StepFrame)])


SIGNATURES["harp_hand_front_fwd"] = (_i, [ctypes.POINTER(HandFront), _vp])
SIGNATURES["harp_hand_front_wide_fwd"] = (_i, [ctypes.POINTER(HandFront), _vp, _vp])
SIGNATURES["harp_hand_front_hybrid_fwd"] = (_i, [ctypes.POINTER(HandFront), _vp])
SIGNATURES["harp_hand_back_bwd"] = (_i, [ctypes.POINTER(HandFront), _vp, _vp, _vp])
SIGNATURES["harp_hand_back_wide_bwd"] = (_i, [ctypes.POINTER(HandFront), _vp, _vp, _vp, _vp])


class ArmFront(ctypes.Structure):
    """mirror of `harp_arm_front` (include/harp_hip.h)"""
    _fields_ = ([("chain", MeshChain), ("tree", TreeModel), ("tables", FrameTables), ("fid", _vp)] +
                [(n, _vp) for n in ("pose_in", "betas", "trans_b", "cam_R", "cam_T", "light_pos", "colors", "lbs_ws", "weights_T")] +
                [("self_shadow", _i), ("step", StepFrame)])


SIGNATURES["harp_arm_front_fwd"] = (_i, [ctypes.POINTER(ArmFront), _vp])
SIGNATURES["harp_arm_back_bwd"] = (_i, [ctypes.POINTER(ArmFront), _vp, _vp, _vp, _vp])
SIGNATURES["harp_arm_front_wide_fwd"] = (_i, [ctypes.POINTER(ArmFront), _vp, _vp])
SIGNATURES["harp_arm_back_wide_bwd"] = (_i, [ctypes.POINTER(ArmFront), _vp, _vp, _vp, _vp, _vp])


class Conv3x3Args(ctypes.Structure):
    """mirror of `harp_conv3x3_args` (include/harp_hip.h)"""
    _fields_ = ([(n, _vp) for n in ("in_", "filters", "bias", "out", "pooled", "target", "target_row", "g_tap", "loss", "gate")] +
                [(n, _i) for n in ("N", "H", "W", "Cin", "Cout", "precision", "epilogue", "in_channels")] + [("tap_scale", _f)] +
                [("tile_list", _vp), ("tile_count", _vp), ("max_tiles", _i), ("in_valid_shift", _i), ("in_valid", _vp), ("in_alt", _vp),
                 ("out_valid", _vp), ("tile_origin", _vp), ("in_valid_origin", _vp), ("out_valid_origin", _vp), ("tile_pitch", _i),
                 ("in_valid_pitch", _i), ("out_valid_pitch", _i), ("tile_side", _i), ("in_valid_cell", _i), ("out_valid_cell", _i)])


SIGNATURES.update({
    "harp_conv3x3_filter_bytes": (_sz, [_i, _i]),
    "harp_conv3x3_pack_filters": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "harp_conv3x3": (_i, [ctypes.POINTER(Conv3x3Args), _vp]),
})


class Vgg16(ctypes.Structure):
    """mirror of `harp_vgg16` (include/harp_hip.h)"""
    _fields_ = [("filters", _vp * 10), ("filters_t", _vp * 10), ("bias", _vp * 10), ("w0t", _vp), ("layer_w", _f * 5), ("precision", _i)]


class Vgg16TermArgs(ctypes.Structure):
    """mirror of `harp_vgg16_term_args` (include/harp_hip.h)"""
    _fields_ = [("rgb", _vp), ("y_true", _vp), ("mask", _vp), ("rows", _vp), ("target", _vp * 4), ("target_by_row", _i), ("covered", _vp),
                ("g_rgb", _vp), ("weight", _f), ("loss", _vp), ("N", _i), ("S", _i), ("ws", _vp), ("target_in", _vp * 10), ("tiles", _vp * 4),
                ("tile_list", _vp * 4), ("tile_count", _vp * 4), ("max_tiles", _i * 4), ("tile_origin", _vp * 4), ("tile_pitch", _i * 4), ("tile_side", _i * 4), ("side_streams", _vp * 3)]


SIGNATURES.update({
    "harp_vgg16_ws_bytes": (_sz, [_i, _i, _i]),
    "harp_vgg16_features": (_i, [ctypes.POINTER(Vgg16), _vp, _vp, _vp, _i, _i, _vp, ctypes.POINTER(_vp), _vp]),
    "harp_vgg16_term": (_i, [ctypes.POINTER(Vgg16), ctypes.POINTER(Vgg16TermArgs), _vp]),
})
