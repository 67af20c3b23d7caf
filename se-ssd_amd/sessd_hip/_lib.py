"""ctypes loader of the gfx950 C-ABI library (include/sessd_hip.h).

The product path has no CPU fallback: if libsessd_hip.so is missing or a symbol is absent this
module raises at import time, and every op raises on a non-zero return code.
"""
import ctypes as C
import os

# torch first, always: torch ships its own libamdhip64.so.7 and this library is linked against the system one of the same SONAME --
# whichever is loaded first serves both. Loading ours first (a test module that imports sessd_hip before torch) put the process on
# the system runtime under torch's other bundled ROCm libraries, and the first kernel launch failed with hipErrorNoDevice.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libsessd_hip.so")

vp, i32, u32, f32, sz, i64, f64 = C.c_void_p, C.c_int, C.c_uint32, C.c_float, C.c_size_t, C.c_longlong, C.c_double



class ChainLevel(C.Structure):
    """sessd_chain_level_t (include/sessd_hip_types.h)"""
    _fields_ = [("ksize", i32 * 3), ("stride", i32 * 3), ("pad", i32 * 3), ("out_dims", i32 * 3), ("cap", i32),
                ("indices", vp), ("n_dev", vp)]


class RulebookJob(C.Structure):
    """sessd_rulebook_job_t (include/sessd_hip_types.h)"""
    _fields_ = [("in_level", i32), ("out_level", i32), ("ksize", i32 * 3), ("stride", i32 * 3), ("pad", i32 * 3),
                ("nbr", vp), ("tile_mask", vp), ("site_mask", vp), ("perm", vp), ("tile_mask_sorted", vp)]


class FillTilesJob(C.Structure):
    """sessd_fill_tiles_job_t (include/sessd_hip_types.h)"""
    _fields_ = [("out", vp), ("value", vp), ("tile_mask", vp), ("cout", i32), ("h", i32), ("w", i32), ("mask_th", i32), ("tile", i32),
                ("near_kind", i32), ("near_mask", vp)]


class HeadLossNet(C.Structure):
    """sessd_head_loss_net_t (include/sessd_hip_types.h)"""
    _fields_ = [("box", vp), ("cls", vp), ("dir", vp), ("iou", vp), ("labels", vp), ("reg_targets", vp), ("anchors", vp)]


class HeadLossCfg(C.Structure):
    """sessd_head_loss_cfg_t (include/sessd_hip_types.h)"""
    _fields_ = [("batch", i32), ("num_anchors", i32), ("labels_i64", i32), ("pos_capacity", i32), ("cons_capacity", i32),
                ("pos_cls_weight", f32), ("neg_cls_weight", f32), ("focal_alpha", f32), ("focal_gamma", f32),
                ("smooth_l1_sigma", f32), ("cls_loss_weight", f32), ("loc_loss_weight", f32), ("dir_loss_weight", f32),
                ("direction_offset", f32), ("score_thresh", f32), ("match_iou_thresh", f32), ("center_range", f32 * 6)]


# name -> (restype, argtypes). Kept in step with include/sessd_hip.h (tests/test_abi.py checks it).
SIGNATURES = {
    "sessd_version": (C.c_char_p, []),
    "sessd_bn_sync_scratch_bytes": (sz, [i32]),
    "sessd_bn_relu_train_stats": (i32, [vp, vp, i32, i32, vp, vp, sz, vp]),
    "sessd_bn_relu_train_apply": (i32, [vp, vp, i32, i32, vp, vp, f32, f32, i32, vp, vp, vp, vp, vp, vp, vp]),
    "sessd_bn_relu_train_bwd_stats": (i32, [vp, vp, vp, vp, i32, i32, vp, vp, i32, vp, vp, vp, vp, sz, vp]),
    "sessd_bn_relu_train_bwd_apply": (i32, [vp, vp, vp, vp, i32, i32, vp, vp, vp, i32, vp, vp, vp, vp, sz, vp]),
    "sessd_bn2d_relu_train_stats": (i32, [vp, i32, i32, i32, vp, vp, sz, vp]),
    "sessd_bn2d_relu_train_apply": (i32, [vp, i32, i32, i32, vp, vp, f32, f32, i32, vp, vp, vp, vp, vp, vp, vp]),
    "sessd_bn2d_relu_train_bwd_stats": (i32, [vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, i32, vp, vp, vp, vp, sz, vp]),
    "sessd_bn2d_relu_train_bwd_apply": (i32, [vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, i32, vp, vp, vp, vp, sz, vp]),
    "sessd_head_loss_workspace_bytes": (sz, [vp]),
    "sessd_head_loss": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    "sessd_fill_u32": (i32, [vp, u32, sz, vp]),
    "sessd_fill_u32_multi": (i32, [i32, vp, vp, vp, vp]),
    "sessd_set_external_clear": (None, [i32]),
    "sessd_stream_create_cu_mask": (i32, [i32, vp, vp]),
    "sessd_stream_destroy": (i32, [vp]),
    "sessd_debug_cu_probe": (i32, [vp, i32, i32, vp]),
    "sessd_hash_capacity": (u32, [i32]),
    "sessd_hash_clear": (i32, [vp, vp, u32, vp]),
    "sessd_voxelize_workspace_bytes": (sz, [u32, i32, i32, i32]),
    "sessd_voxelize_frame": (i32, [vp, i32, i32, vp, vp, vp, i32, i32, i32, vp, vp, u32, vp, vp, i32, vp, vp, vp, vp, sz, vp]),
    "sessd_stage_points": (i32, [vp, i32, vp, i32, vp]),
    "sessd_voxelize_frames_workspace_bytes": (sz, [u32, i32, i32, i32, i32]),
    "sessd_voxelize_frames": (i32, [vp, i32, i32, i32, vp, vp, vp, i32, i32, vp, vp, u32, vp, vp, i32, vp, vp, vp, vp, sz, vp]),
    "sessd_vfe_mean": (i32, [vp, vp, vp, i32, i32, i32, i32, vp, vp]),
    "sessd_boxes_pairwise": (i32, [i32, vp, i32, vp, i32, vp, vp]),
    "sessd_boxes_aligned_overlap_bev": (i32, [vp, vp, i32, vp, vp]),
    "sessd_rotate_iou_eval": (i32, [vp, i32, vp, i32, i32, vp, vp]),
    "sessd_box3d_overlap_eval": (i32, [vp, i32, vp, i32, i32, i32, f64, vp, vp]),
    "sessd_kitti_statistics": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f64, vp, i32, i32, i32, vp, vp, vp, vp]),
    "sessd_kitti_reduce": (i32, [vp, i32, i32, vp, vp]),
    "sessd_kitti_thresholds": (i32, [vp, i32, i32, i32, vp, vp, vp]),
    "sessd_nms_workspace_bytes": (sz, [i32]),
    "sessd_nms_sorted": (i32, [i32, vp, i32, f32, vp, vp, vp, sz, vp]),
    "sessd_nms_axis_eps_sorted": (i32, [vp, i32, i32, f32, f32, vp, vp, vp, sz, vp]),
    "sessd_conv2d_mfma": (i32, [vp, i32, i32, i32, i32, vp, i32, vp, vp, i32, i32, i32, vp, i32, i32, i32, i32, i32, i32,
                                vp, vp, i32, vp, i32, vp]),
    "sessd_conv3x3_winograd": (i32, [vp, i32, i32, i32, i32, vp, vp, i32, vp, vp, i32, vp, i32, vp]),
    "sessd_conv2d_pack_taps": (i32, [vp, i64, i64, vp, i32, i32, i32, vp, vp]),
    "sessd_conv3x3_winograd_pack": (i32, [vp, i64, i64, i32, i32, i32, i32, vp, vp]),
    "sessd_conv3x3_winograd_sk_workspace_bytes": (sz, [i32, i32, i32, i32, i32, i32]),
    "sessd_conv3x3_winograd_sk_sets": (i32, [vp, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp, i32, vp, vp, sz, i32, i32, vp]),
    "sessd_conv3x3_winograd_sk": (i32, [vp, i32, i32, i32, i32, vp, vp, i32, vp, vp, i32, vp, vp, sz, i32, i32, vp]),
    "sessd_conv3x3_winograd_sk_active": (i32, [vp, i32, i32, i32, i32, vp, vp, i32, vp, vp, i32, vp, vp, vp, i32, i32, vp, sz, i32, i32, vp]),
    "sessd_bev_tile_activity_workspace_bytes": (sz, [i32, i32]),
    "sessd_bev_tile_activity": (i32, [vp, vp, i32, i32, i32, i32, vp, i32, vp, vp, vp, i32, vp, sz, vp]),
    "sessd_fill_inactive_tiles": (i32, [vp, i32, i32, vp]),
    "sessd_conv2d_sk_workspace_bytes": (sz, [i32, i32, i32, i32, i32, i32]),
    "sessd_conv2d_sk_pack": (i32, [vp, i64, i64, vp, i32, i32, i32, vp, vp]),
    "sessd_conv2d_sk": (i32, [vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, i32, i32, i32, vp, i32, i32, i32, i32, vp, vp, vp, vp,
                              i32, vp, vp, sz, i32, vp]),
    "sessd_conv2d_sk_active": (i32, [vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, i32, i32, i32, vp, i32, i32, i32, i32, vp, vp, vp, vp,
                                     i32, vp, vp, vp, i32, i32, vp, sz, i32, vp]),
    "sessd_conv2d_mfma_active": (i32, [vp, i32, i32, i32, i32, vp, i32, vp, vp, i32, i32, i32, vp, i32, i32, i32, i32, i32, i32, vp, vp, i32, vp,
                                       vp, vp, i32, i32, vp]),
    "sessd_deconv2d_s2_mfma_pair_active": (i32, [vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, i32, vp, vp, vp, vp,
                                                 i32, i32, vp]),
    "sessd_deconv2d_s2_mfma_pair": (i32, [vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, i32, vp, vp, i32, vp]),
    "sessd_deconv2d_s2_mfma": (i32, [vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, i32, vp, vp, i32, vp, i32, vp]),
    "sessd_ssfa_fuse": (i32, [vp, vp, vp, vp, f32, f32, f32, f32, i32, i32, i32, vp, vp]),
    "sessd_ssfa_fuse_head": (i32, [vp, vp, vp, vp, f32, f32, f32, f32, i32, i32, i32, vp, vp, vp, i32, vp, vp]),
    "sessd_predict_workspace_bytes": (sz, [i32, i32, i32, i32]),
    "sessd_predict": (i32, [vp, i32, i32, vp, i32, vp, f32, i32, i32, f32, vp, f32, vp, vp, vp, vp, vp, sz, vp]),
    "sessd_predict_fused": (i32, [vp, i32, i32, vp, i32, vp, f32, i32, i32, f32, vp, f32, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, sz, vp]),
    "sessd_ssfa_fuse_head_keys": (i32, [vp, vp, vp, vp, f32, f32, f32, f32, i32, i32, i32, vp, vp, vp, i32, vp, f32, vp, i32, vp, vp]),
    "sessd_di_nms_workspace_bytes": (sz, [i32]),
    "sessd_di_nms": (i32, [vp, vp, vp, i32, vp, vp, vp, vp, vp, i32, f32, vp, i32, vp, f32, i32, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    "sessd_pack_detections": (i32, [vp, vp, vp, vp, i32, i32, vp, vp, i32, vp, vp]),
    "sessd_quads_pairwise": (i32, [i32, vp, i32, vp, i32, vp, f32, vp, vp]),
    "sessd_rotate_nms_workspace_bytes": (sz, [i32]),
    "sessd_rotate_nms_sorted": (i32, [vp, i32, f32, i32, vp, vp, vp, sz, vp]),
    "sessd_rotate_nms_corners_sorted": (i32, [vp, i32, f32, i32, vp, vp, vp, sz, vp]),
    "sessd_sparse_hash_build": (i32, [vp, vp, i32, vp, vp, vp, u32, vp]),
    "sessd_sparse_downsample_workspace_bytes": (sz, [i32, i32, u32]),
    "sessd_sparse_downsample_sites": (i32, [vp, vp, i32, vp, vp, vp, vp, vp, vp, u32, vp, i32, vp, vp, vp, sz, vp]),
    "sessd_sparse_rulebook": (i32, [vp, vp, i32, vp, vp, vp, vp, vp, u32, vp, vp, vp, vp]),
    "sessd_sparse_rulebook_pair": (i32, [vp, vp, i32, vp, vp, vp, vp, vp, u32, vp, vp, vp, vp, vp, vp, vp, vp, u32, vp, vp, vp, vp]),
    "sessd_sparse_downsample_sites_unordered": (i32, [vp, vp, i32, vp, vp, vp, vp, vp, vp, u32, vp, i32, vp, vp, vp]),
    "sessd_sparse_to_dense": (i32, [vp, vp, i32, i32, vp, vp, vp]),
    "sessd_dense_to_sparse": (i32, [vp, vp, i32, i32, vp, vp, vp]),
    "sessd_sparse_to_dense_dev": (i32, [vp, vp, i32, vp, i32, vp, vp, vp]),
    "sessd_dense_to_sparse_dev": (i32, [vp, vp, i32, vp, i32, vp, vp, vp]),
    "sessd_sparse_chain_workspace_bytes": (sz, [i32, i32, vp]),
    "sessd_sparse_chain_sites": (i32, [vp, vp, i32, i32, i32, vp, vp, sz, i32, vp, vp]),
    "sessd_sparse_chain_rulebooks": (i32, [vp, vp, i32, vp, vp, u32, vp, i32, i32, vp, vp, i32, vp, vp]),
    "sessd_sparse_pack_weight": (i32, [vp, i32, i32, i32, vp, vp]),
    "sessd_sparse_pack_weight_adjoint": (i32, [vp, i32, i32, i32, i32, vp, vp]),
    "sessd_sparse_conv": (i32, [vp, i32, vp, vp, i32, vp, i32, vp, vp, vp, i32, vp, i32, vp, vp, vp, i32, vp]),
    "sessd_sparse_conv_sorted": (i32, [vp, i32, vp, vp, i32, vp, i32, vp, vp, vp, i32, vp, i32, vp, vp, vp, i32, vp, vp]),
    "sessd_sparse_renumber_workspace_bytes": (sz, [i32, vp]),
    "sessd_sparse_renumber_sites": (i32, [vp, vp, i32, i32, vp, vp, i32, vp, vp, u32, vp, vp, vp, sz, vp]),
    "sessd_points_in_bodies": (i32, [vp, i32, i32, vp, i32, i32, vp, vp]),
    "sessd_points_rigid_moves": (i32, [vp, i32, i32, vp, vp, vp, vp, vp, i32, vp]),
    "sessd_points_global_transform": (i32, [vp, i32, i32, i32, f32, f32, f32, vp, vp]),
    "sessd_points_compact_workspace_bytes": (sz, [i32]),
    "sessd_points_compact": (i32, [vp, vp, i32, i32, vp, i32, vp, vp, sz, vp]),
    "sessd_farthest_point_sample": (i32, [vp, i32, i32, i32, vp, vp]),
    "sessd_bn_relu_train_workspace_bytes": (sz, [i32]),
    "sessd_bn_relu_train_fwd": (i32, [vp, vp, i32, i32, vp, vp, f32, f32, i32, vp, vp, vp, vp, vp, vp, sz, vp]),
    "sessd_bn_relu_train_bwd": (i32, [vp, vp, vp, vp, i32, i32, vp, vp, vp, i32, vp, vp, vp, vp, sz, vp]),
    "sessd_bn2d_relu_train_workspace_bytes": (sz, [i32]),
    "sessd_bn2d_relu_train_fwd": (i32, [vp, i32, i32, i32, vp, vp, f32, f32, i32, vp, vp, vp, vp, vp, vp, sz, vp]),
    "sessd_bn2d_relu_train_bwd": (i32, [vp, vp, vp, i32, i32, i32, vp, vp, vp, i32, vp, vp, vp, vp, sz, vp]),
    "sessd_bn2d_relu_train_bwd_x": (i32, [vp, vp, i32, i32, i32, vp, vp, vp, vp, i32, vp, vp, vp, vp, sz, vp]),
    "sessd_sparse_rulebook_transpose": (i32, [vp, i32, vp, i32, i32, vp, vp, vp]),
    "sessd_sparse_conv_wgrad_workspace_bytes": (sz, [i32, i32, i32]),
    "sessd_sparse_conv_wgrad": (i32, [vp, i32, vp, i32, vp, vp, i32, vp, i32, vp, vp, sz, vp]),
    "sessd_grad_clip_workspace_bytes": (sz, []),
    "sessd_grad_clip_coef": (i32, [vp, sz, f32, vp, sz, vp, vp]),
    "sessd_conv2d_wgrad_workspace_bytes": (sz, [i32, i32, i32]),
    "sessd_conv2d_wgrad": (i32, [vp, i32, i32, i32, i32, vp, i32, i32, i32, i32, i32, vp, vp, sz, vp]),
    "sessd_assign_targets_workspace_bytes": (sz, [i32]),
    "sessd_assign_targets": (i32, [vp, i32, vp, vp, i32, f32, f32, vp, vp, vp, vp, vp, sz, vp]),
    "sessd_odiou3d": (i32, [vp, vp, i32, vp, vp, vp]),
    "sessd_adam_ema_step": (i32, [vp, vp, vp, vp, vp, sz, f64, f64, f64, f64, f64, i32, vp, f64, vp]),
    "sessd_one_cycle_args": (i32, [vp, i32, f64, f64, f64, f64, f64, f64, f64, f64, vp, vp, vp]),
    "sessd_adam_ema_step_dev": (i32, [vp, vp, vp, vp, vp, sz, vp, vp, vp]),
    "sessd_sum_f32": (i32, [vp, sz, f32, vp, sz, vp, vp]),
    "sessd_nchw_channel_sum": (i32, [vp, i32, i32, i32, vp, vp, sz, vp]),
    "sessd_nchw_split_nhwc": (i32, [vp, i32, i32, i32, i32, vp, vp, vp]),
    "sessd_nhwc_merge_nchw": (i32, [vp, i32, vp, i32, i32, i32, vp, vp]),
    "sessd_box_collision_host": (i32, [vp, i32, vp, i32, i32, i32, vp]),
    "sessd_noise_per_box_host": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "sessd_dense_pack_batch": (i32, [vp, i32, i32, vp]),
    "sessd_sparse_pack_batch": (i32, [vp, i32, i32, vp]),
    "sessd_conv3x3_wgrad_winograd_workspace_bytes": (sz, [i32, i32, i32, i32, i32]),
    "sessd_conv3x3_wgrad_winograd": (i32, [vp, i32, i32, i32, i32, vp, i32, vp, vp, sz, vp]),
    "sessd_ssfa_fuse_train_workspace_bytes": (sz, [i32, i32, i32]),
    "sessd_ssfa_fuse_train_fwd": (i32, [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, f32, f32, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    "sessd_ssfa_fuse_train_bwd": (i32, [vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
}


class SessdError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libsessd_hip.so not built (%s). Run `python se-ssd_amd/build.py` (hipcc --offload-arch=gfx950); "
            "there is no CPU fallback for the product path." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(rc, what):
    if rc != 0:
        kind = {-1: "invalid argument", -2: "workspace too small"}.get(rc, "hipError_t %d" % rc)
        raise SessdError("%s failed: %s" % (what, kind))
