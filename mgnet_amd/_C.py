"""ctypes binding of include/mgnet_hip.h (libmgnet_hip.so).  Fails loudly when the extension is missing."""
import ctypes
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MGNET_HIP_LIB", os.path.join(_HERE, "lib", "libmgnet_hip.so"))  # env override: A/B builds

MGN_MAX_SCALES = 4
_ERR = {-22: "MGN_EINVAL (bad shape / null pointer)", -28: "MGN_ENOSPC (workspace too small)",
        -95: "MGN_ENOTSUP (option has no kernel)", -5: "MGN_ELAUNCH (kernel launch failed)"}

# every symbol include/mgnet_hip.h declares (tests check that the library exports all of them)
SYMBOLS = ["mgn_version", "mgn_reproj_workspace_bytes", "mgn_reproj_loss_fwd", "mgn_reproj_loss_bwd",
           "mgn_iabn_workspace_bytes", "mgn_iabn_stats", "mgn_iabn_train_coeffs", "mgn_iabn_combine", "mgn_iabn_eval_coeffs", "mgn_iabn_apply",
           "mgn_iabn_bwd_reduce", "mgn_iabn_bwd_apply",
           "mgn_optim_chunk", "mgn_sqnorm", "mgn_clip_coef", "mgn_adam_step", "mgn_adam_step_dev", "mgn_optim_step_dev", "mgn_clip_coef_scaled", "mgn_conv_igemm", "mgn_conv_igemm_stats", "mgn_conv_stat_rows", "mgn_conv3x3_win", "mgn_conv3x3_up2_win", "mgn_conv_win_patch_rows", "mgn_conv_stem7", "mgn_conv_stem7_blocks", "mgn_iabn_coeffs_from_partials", "mgn_iabn_partials_reduce", "mgn_conv_wgrad", "mgn_conv_wgrad_partial", "mgn_conv1x1_cat", "mgn_conv1x1_split", "mgn_conv_wgrad_cat", "mgn_conv3x3_win_act", "mgn_conv_igemm_act", "mgn_conv_wgrad_reduce_batch", "mgn_conv_wgrad_reduce_blocks", "mgn_conv_wgrad_workspace_bytes", "mgn_weight_layout", "mgn_weight_layout_batch",
           "mgn_upce_partials", "mgn_adjoint_footprint_floats", "mgn_upce_fwd", "mgn_upce_bwd", "mgn_ohem_select_workspace_bytes", "mgn_ohem_select", "mgn_ins_loss_fwd", "mgn_ins_loss_bwd", "mgn_prep_input",
           "mgn_upsample1_fwd", "mgn_upsample1_bwd", "mgn_maxpool3x3s2_fwd", "mgn_maxpool3x3s2_bwd",
           "mgn_add_relu_fwd", "mgn_sum3", "mgn_relu_mask_bwd", "mgn_colsum", "mgn_bcast_rows", "mgn_scale_channels", "mgn_nearest_fwd",
           "mgn_nearest_bwd", "mgn_concat2", "mgn_split2", "mgn_vec_linear_fwd", "mgn_vec_linear_bwd_workspace_bytes", "mgn_vec_linear_bwd",
           "mgn_panoptic_targets_workspace_bytes", "mgn_panoptic_targets",
           "mgn_panoptic_post_workspace_bytes", "mgn_panoptic_post", "mgn_instance_post_workspace_bytes", "mgn_instance_post", "mgn_instance_masks", "mgn_pseudo_label_ids", "mgn_depth_post_workspace_bytes", "mgn_depth_post",
           "mgn_depth_metrics_workspace_bytes", "mgn_depth_metrics", "mgn_abn_maxpool_fwd", "mgn_abn_maxpool_bwd",
           "mgn_iabn_bwd_reduce_x", "mgn_iabn_bwd_reduce_x_relu", "mgn_iabn_bwd_apply_x", "mgn_abn_add_relu_fwd", "mgn_u8_frames_to_f32", "mgn_u8_frames_to_f32_nhwc4", "mgn_u8_frames_to_rgbx", "mgn_msc_input", "mgn_msc_accumulate", "mgn_uncertainty_fwd", "mgn_uncertainty_bwd", "mgn_copy_from_host", "mgn_head_act_fwd", "mgn_head_act_bwd",
           "mgn_p2p_mailbox_bytes", "mgn_p2p_alloc", "mgn_p2p_free", "mgn_p2p_export", "mgn_p2p_open", "mgn_p2p_close", "mgn_p2p_exchange",
           "mgn_geometry_partial_rows", "mgn_view_synthesis_fwd", "mgn_view_synthesis_bwd", "mgn_reconstruct_fwd",
           "mgn_reconstruct_bwd", "mgn_project_fwd", "mgn_project_bwd",
           "mgn_plan_begin", "mgn_plan_recorded", "mgn_plan_current", "mgn_plan_end", "mgn_plan_abort", "mgn_plan_node_count", "mgn_plan_node_info", "mgn_plan_node_args",
           "mgn_plan_compile", "mgn_plan_set_stream", "mgn_plan_run", "mgn_plan_prof_elapsed", "mgn_plan_free",
           "mgn_plan_trace", "mgn_plan_trace_read", "mgn_plan_set_skip", "mgn_plan_node_ro", "mgn_plan_node_ro_family", "mgn_plan_set_jitter", "mgn_plan_probe",
           "mgn_abn_apply_pool", "mgn_att_abn_bwd_stats", "mgn_att_abn_bwd_sums", "mgn_att_abn_bwd_apply"]
SYMBOLS_F16 = [n + "_f16" for n in ['mgn_weight_layout', 'mgn_weight_layout_batch', 'mgn_conv_igemm', 'mgn_conv_igemm_stats', 'mgn_conv3x3_win', 'mgn_conv3x3_up2_win', 'mgn_conv_stem7', 'mgn_conv_wgrad', 'mgn_conv_wgrad_partial', 'mgn_conv1x1_cat', 'mgn_conv1x1_split', 'mgn_conv_wgrad_cat', 'mgn_conv3x3_win_act', 'mgn_conv_igemm_act', 'mgn_add_relu_fwd', 'mgn_sum3', 'mgn_abn_add_relu_fwd', 'mgn_relu_mask_bwd', 'mgn_colsum', 'mgn_bcast_rows', 'mgn_scale_channels', 'mgn_nearest_fwd', 'mgn_nearest_bwd', 'mgn_abn_maxpool_fwd', 'mgn_abn_maxpool_bwd', 'mgn_maxpool3x3s2_fwd', 'mgn_maxpool3x3s2_bwd', 'mgn_upce_fwd', 'mgn_upce_bwd', 'mgn_ins_loss_fwd', 'mgn_ins_loss_bwd', 'mgn_prep_input', 'mgn_iabn_stats', 'mgn_iabn_train_coeffs', 'mgn_iabn_apply', 'mgn_iabn_bwd_reduce', 'mgn_iabn_bwd_reduce_x', 'mgn_iabn_bwd_reduce_x_relu', 'mgn_iabn_bwd_apply', 'mgn_iabn_bwd_apply_x', 'mgn_abn_apply_pool', 'mgn_att_abn_bwd_stats', 'mgn_att_abn_bwd_apply']]
DEPTH_MAX_FILTER_IDS = 16
MGN_MAX_TASKS = 8   # include/mgnet_hip.h


class DepthPostCfg(ctypes.Structure):   # mgn_depth_post_cfg
    _fields_ = [("H", ctypes.c_int), ("W", ctypes.c_int), ("use_dgc_scaling", ctypes.c_int), ("has_panoptic", ctypes.c_int),
                ("fx", ctypes.c_float), ("fy", ctypes.c_float), ("cx", ctypes.c_float), ("cy", ctypes.c_float),
                ("real_camera_height", ctypes.c_float), ("n_filter", ctypes.c_int), ("road_class_id", ctypes.c_int64),
                ("filter_ids", ctypes.c_int64 * 16)]
PANOPTIC_MAX_CENTERS = 65534   # MGN_PANOPTIC_MAX_CENTERS


class PanopticCfg(ctypes.Structure):   # mgn_panoptic_cfg
    _fields_ = [(n, ctypes.c_int) for n in ("H", "W", "num_thing_classes", "last_stuff_id", "label_divisor", "stuff_area",
                                            "void_label")] + [("threshold", ctypes.c_float), ("nms_kernel", ctypes.c_int)]
TARGETS_MAX_SEGMENTS = 1024   # MGN_TARGETS_MAX_SEGMENTS


class TargetsCfg(ctypes.Structure):   # mgn_targets_cfg
    _fields_ = [(n, ctypes.c_int) for n in ("B", "H", "W", "pan_rgb", "ignore_label", "sigma", "first_thing_id",
                                            "ignore_stuff_in_offset", "small_instance_area", "small_instance_weight",
                                            "ignore_crowd_in_semantic", "legacy_promotion", "max_segments")] + \
               [("depth_ignore_mask", ctypes.c_uint32 * 8)]


class PlanNodeInfo(ctypes.Structure):   # mgn_plan_node_info_t
    _fields_ = [("type", ctypes.c_int), ("which", ctypes.c_int), ("stream", ctypes.c_void_p), ("func", ctypes.c_void_p),
                ("name", ctypes.c_char_p), ("grid", ctypes.c_uint * 3), ("block", ctypes.c_uint * 3), ("shmem", ctypes.c_size_t),
                ("nargs", ctypes.c_int), ("nbytes", ctypes.c_int), ("blob", ctypes.c_void_p)]


class ReprojCfg(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int), ("n_scales", ctypes.c_int),
                ("ssim_loss_weight", ctypes.c_float), ("photometric_loss_weight", ctypes.c_float),
                ("smoothing_loss_weight", ctypes.c_float), ("automask_loss", ctypes.c_int),
                ("photometric_reduce_op", ctypes.c_int), ("padding_mode", ctypes.c_int), ("rows_per_wave", ctypes.c_int),
                ("frame_layout", ctypes.c_int), ("prof_begin", ctypes.c_void_p), ("prof_end", ctypes.c_void_p)]


_lib = None
PLAN_RECORDER = [None]   # engine/plan.py: the recorder of the step being recorded (launch-plan replay), else None


def plan_touch(reads=(), writes=()):
    """tell a running plan recording which tensors the NEXT library call reaches through pointers that live in device memory
    (descriptor tables) rather than in its arguments; no-op otherwise"""
    rec = PLAN_RECORDER[0]
    if rec is not None:
        rec.touch(list(reads), list(writes))


H16 = (torch.bfloat16, torch.float16)   # the 16-bit activation formats: bf16 entry points, or their _f16 twins (csrc/h16.h)
F16_TWINS = ['mgn_weight_layout', 'mgn_weight_layout_batch', 'mgn_conv_igemm', 'mgn_conv_igemm_stats', 'mgn_conv3x3_win', 'mgn_conv3x3_up2_win', 'mgn_conv_stem7', 'mgn_conv_wgrad', 'mgn_conv_wgrad_partial', 'mgn_conv1x1_cat', 'mgn_conv1x1_split', 'mgn_conv_wgrad_cat', 'mgn_conv3x3_win_act', 'mgn_conv_igemm_act', 'mgn_add_relu_fwd', 'mgn_sum3', 'mgn_abn_add_relu_fwd', 'mgn_relu_mask_bwd', 'mgn_colsum', 'mgn_bcast_rows', 'mgn_scale_channels', 'mgn_nearest_fwd', 'mgn_nearest_bwd', 'mgn_abn_maxpool_fwd', 'mgn_abn_maxpool_bwd', 'mgn_maxpool3x3s2_fwd', 'mgn_maxpool3x3s2_bwd', 'mgn_upce_fwd', 'mgn_upce_bwd', 'mgn_ins_loss_fwd', 'mgn_ins_loss_bwd', 'mgn_prep_input', 'mgn_iabn_stats', 'mgn_iabn_train_coeffs', 'mgn_iabn_apply', 'mgn_iabn_bwd_reduce', 'mgn_iabn_bwd_reduce_x', 'mgn_iabn_bwd_reduce_x_relu', 'mgn_iabn_bwd_apply', 'mgn_iabn_bwd_apply_x', 'mgn_abn_apply_pool', 'mgn_att_abn_bwd_stats', 'mgn_att_abn_bwd_apply']


def _fn(name, t):
    """entry point `name` for the activation format of tensor `t` (or of a dtype): fp16 -> the `_f16` twin"""
    dt = t if isinstance(t, torch.dtype) else t.dtype
    return getattr(lib(), name + "_f16") if dt == torch.float16 else getattr(lib(), name)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: the HIP extension is not built. Run `python -m mgnet_amd.build` "
                "(needs hipcc). mgnet_amd has no CPU fallback by design.")
        L = ctypes.CDLL(LIB_PATH)
        vp, ci, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
        L.mgn_version.restype = ctypes.c_char_p
        L.mgn_reproj_workspace_bytes.restype = ci
        L.mgn_reproj_workspace_bytes.argtypes = [ctypes.POINTER(ReprojCfg), ctypes.POINTER(sz)]
        L.mgn_reproj_loss_fwd.restype = ci
        L.mgn_reproj_loss_fwd.argtypes = [ctypes.POINTER(ReprojCfg), vp, vp, vp, vp, vp, vp, ci, ci, vp, ci, vp, vp, vp,
                                          vp, vp, sz, vp]
        L.mgn_reproj_loss_bwd.restype = ci
        L.mgn_reproj_loss_bwd.argtypes = [ctypes.POINTER(ReprojCfg), vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]
        cl, cf = ctypes.c_long, ctypes.c_float
        L.mgn_iabn_workspace_bytes.argtypes = [cl, ci, ci, ctypes.POINTER(sz)]
        L.mgn_iabn_stats.argtypes = [vp, ci, cl, ci, vp, vp, sz, vp]
        L.mgn_iabn_train_coeffs.argtypes = [vp, ci, cl, ci, vp, vp, cf, cf, vp, vp, vp, vp, sz, vp]
        L.mgn_iabn_combine.argtypes = [vp, ci, ci, vp, vp, cf, cf, vp, vp, vp, vp, vp, vp]
        L.mgn_iabn_eval_coeffs.argtypes = [ci, vp, vp, vp, vp, cf, vp, vp, vp]
        L.mgn_iabn_apply.argtypes = [vp, vp, ci, cl, ci, vp, vp, ci, cf, vp]
        L.mgn_iabn_bwd_reduce.argtypes = [vp, vp, ci, cl, ci, vp, vp, cf, ci, cf, vp, vp, vp, sz, vp]
        L.mgn_iabn_bwd_apply.argtypes = [vp, vp, vp, ci, cl, ci, vp, vp, vp, vp, cf, cf, ci, cf, vp]
        L.mgn_sqnorm.argtypes = [vp, cl, vp, ci, ctypes.POINTER(ci), vp]
        L.mgn_clip_coef.argtypes = [vp, ci, cf, cf, vp, vp]
        L.mgn_adam_step.argtypes = [vp, vp, vp, vp, cl, vp, vp, cf, cf, cf, ci, vp, cf, vp]
        L.mgn_adam_step_dev.argtypes = [vp, vp, vp, vp, cl, vp, vp, cf, cf, cf, vp, vp, cf, vp]
        L.mgn_optim_step_dev.argtypes = [ci, vp, vp, vp, vp, cl, vp, vp, cf, cf, cf, ci, vp, vp, cf, vp]
        L.mgn_clip_coef_scaled.argtypes = [vp, ci, cf, cf, cf, cf, ci, vp, vp, vp, vp]
        L.mgn_conv_igemm.argtypes = [vp, vp, vp, vp] + [ci] * 14 + [vp, vp]
        L.mgn_conv3x3_win.argtypes = [vp, vp, vp] + [ci] * 5 + [vp, ci, vp, vp, vp]
        L.mgn_conv3x3_up2_win.argtypes = [vp, vp, vp] + [ci] * 8 + [vp, ci, vp]
        L.mgn_conv_stem7.argtypes = [vp, vp, vp] + [ci] * 7 + [vp, vp]
        L.mgn_conv_stem7_blocks.argtypes = [ci] * 7
        L.mgn_conv_win_patch_rows.argtypes = [ci] * 5
        L.mgn_conv_stat_rows.argtypes = [ci] * 11 + [ctypes.POINTER(ci)]
        L.mgn_conv_igemm_stats.argtypes = [vp, vp, vp] + [ci] * 11 + [vp, vp, vp]
        L.mgn_iabn_coeffs_from_partials.argtypes = [vp, ci, ci, cl, vp, vp, vp, cf, cf, vp, vp, vp, vp, vp]
        L.mgn_iabn_partials_reduce.argtypes = [vp, ci, ci, ci, vp, vp]
        L.mgn_conv_wgrad.argtypes = [vp, vp, vp] + [ci] * 12 + [vp, sz, vp]
        L.mgn_conv_wgrad_workspace_bytes.argtypes = [ci] * 7 + [ctypes.POINTER(sz)]
        for sfx in ("", "_f16"):
            getattr(L, "mgn_conv_igemm_act" + sfx).argtypes = [vp, vp, vp, vp] + [ci] * 12 + [cf, vp, vp]
            getattr(L, "mgn_conv3x3_win_act" + sfx).argtypes = [vp, vp, vp] + [ci] * 5 + [vp, ci, vp, vp, vp, ci, cf, vp]
            getattr(L, "mgn_conv1x1_cat" + sfx).argtypes = [vp, vp, vp, vp] + [ci] * 5 + [vp]
            getattr(L, "mgn_conv1x1_split" + sfx).argtypes = [vp, vp, vp, vp] + [ci] * 5 + [vp]
            getattr(L, "mgn_conv_wgrad_cat" + sfx).argtypes = [vp, vp, vp, vp] + [ci] * 5 + [vp, sz, vp, vp]
        L.mgn_conv_wgrad_partial.argtypes = [vp, vp] + [ci] * 12 + [vp, sz, ctypes.POINTER(ctypes.c_longlong), vp]
        L.mgn_conv_wgrad_reduce_batch.argtypes = [vp, ci, cl, vp]
        L.mgn_conv_wgrad_reduce_blocks.argtypes = [ci, ci, ci]
        L.mgn_weight_layout.argtypes = [vp, vp, ci, ci, ci, ci, ci, ci, ci, vp]
        L.mgn_weight_layout_batch.argtypes = [vp, ci, cl, vp]
        L.mgn_upce_partials.argtypes = [ci, ci, ci]
        L.mgn_upce_fwd.argtypes = [vp, cl, cl, cl, ci, ci, ci, ci, ci, ci, vp, vp, ci, cf, vp, vp, vp, vp]
        L.mgn_upce_bwd.argtypes = [vp, cl, cl, cl, ci, ci, ci, ci, ci, ci, ci, vp, vp, ci, vp, vp, vp, vp, vp, vp]
        L.mgn_adjoint_footprint_floats.argtypes = [ci] * 7 + [vp]
        L.mgn_ohem_select_workspace_bytes.argtypes = [cl, ctypes.POINTER(sz)]
        L.mgn_ohem_select.argtypes = [vp, cl, vp, cf, cl, ci, vp, vp, vp, sz, vp]
        L.mgn_ins_loss_fwd.argtypes = [vp, cl, cl, cl, vp, cl, cl, cl, ci, ci, ci, ci, ci, vp, vp, vp, vp, cf, vp, vp, vp]
        L.mgn_ins_loss_bwd.argtypes = [vp, cl, cl, cl, vp, cl, cl, cl, ci, ci, ci, ci, ci, vp, vp, vp, vp, cf, vp, vp, vp, vp, vp]
        L.mgn_upsample1_fwd.argtypes = [vp, ci, ci, ci, ci, ci, vp, vp]
        L.mgn_upsample1_bwd.argtypes = [vp, ci, ci, ci, ci, ci, vp, vp, vp]
        L.mgn_maxpool3x3s2_fwd.argtypes = [vp, vp, vp, ci, ci, ci, ci, vp]
        L.mgn_maxpool3x3s2_bwd.argtypes = [vp, vp, vp, ci, ci, ci, ci, vp]
        L.mgn_add_relu_fwd.argtypes = [vp, vp, vp, cl, vp]
        L.mgn_sum3.argtypes = [vp, vp, vp, vp, cl, vp]
        L.mgn_relu_mask_bwd.argtypes = [vp, vp, vp, cl, vp]
        L.mgn_colsum.argtypes = [vp, vp, ci, cl, ci, cf, vp, vp, sz, vp]
        L.mgn_bcast_rows.argtypes = [vp, ci, cl, ci, cf, vp, vp]
        L.mgn_scale_channels.argtypes = [vp, vp, ci, cl, ci, ci, vp, vp, vp, vp]
        L.mgn_vec_linear_fwd.argtypes = [vp, vp, ci, ci, ci, ci, vp, vp, vp, vp, ci, cf, cf, vp, vp, vp, vp]
        L.mgn_vec_linear_bwd_workspace_bytes.argtypes = [ci, ci, ci, ctypes.POINTER(sz)]
        L.mgn_vec_linear_bwd.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, vp, vp, vp, cf, cf, vp, vp, vp, vp, vp, sz, vp]
        L.mgn_nearest_fwd.argtypes = [vp, ci, ci, ci, ci, ci, ci, vp, vp]
        L.mgn_nearest_bwd.argtypes = [vp, ci, ci, ci, ci, ci, ci, vp, vp]
        L.mgn_concat2.argtypes = [vp, vp, cl, ci, ci, vp, vp]
        L.mgn_split2.argtypes = [vp, cl, ci, ci, vp, vp, vp]
        L.mgn_prep_input.argtypes = [vp, ci, ci, ci, ci, vp, vp, vp, ci, vp]
        L.mgn_u8_frames_to_f32.argtypes = [vp, ci, cl, cf, vp, vp]
        L.mgn_u8_frames_to_f32_nhwc4.argtypes = [vp, ci, cl, cf, vp, vp]
        L.mgn_u8_frames_to_rgbx.argtypes = [vp, ci, cl, vp, vp]
        L.mgn_uncertainty_fwd.argtypes = [vp, ci, vp, ctypes.c_uint, vp, vp, vp]
        L.mgn_uncertainty_bwd.argtypes = [vp, vp, ci, ci, vp, ctypes.c_uint, vp, vp, vp]
        L.mgn_copy_from_host.argtypes = [vp, vp, sz, vp]
        L.mgn_head_act_fwd.argtypes = [vp, ci, ci, ci, ci, ci, ci, ci, vp, vp]
        L.mgn_head_act_bwd.argtypes = [vp, cl, cl, cl, vp, ci, ci, ci, ci, ci, ci, ci, cf, vp, vp]
        L.mgn_p2p_mailbox_bytes.argtypes = []
        L.mgn_p2p_alloc.argtypes = [ctypes.POINTER(vp)]
        L.mgn_p2p_free.argtypes = [vp]
        L.mgn_p2p_export.argtypes = [vp, vp]
        L.mgn_p2p_open.argtypes = [vp, ctypes.POINTER(vp)]
        L.mgn_p2p_close.argtypes = [vp]
        L.mgn_p2p_exchange.argtypes = [vp, ci, ci, ci, ctypes.c_uint, vp, vp, ci, ci, vp, vp, cf, vp]
        L.mgn_msc_input.argtypes = [vp, ci, ci, ci, ci, ci, ci, ci, vp, vp]
        L.mgn_msc_accumulate.argtypes = [vp, ci, cl, cl, cl, cl] + [ci] * 9 + [cf, cf, cf, vp, vp]
        L.mgn_iabn_bwd_reduce_x.argtypes = [vp, vp, ci, cl, ci, vp, vp, vp, vp, cf, ci, cf, vp, vp, vp, sz, vp]
        L.mgn_iabn_bwd_reduce_x_relu.argtypes = [vp, vp, vp, vp, vp, cl, ci, vp, vp, vp, vp, cf, vp, vp, vp, sz, vp]
        L.mgn_iabn_bwd_apply_x.argtypes = [vp, vp, vp, ci, cl, ci, vp, vp, vp, vp, vp, vp, cf, cf, ci, cf, vp]
        L.mgn_abn_add_relu_fwd.argtypes = [vp, vp, vp, vp, vp, vp, cl, ci, vp]
        L.mgn_abn_maxpool_fwd.argtypes = [vp, vp, vp, ci, cf, vp, vp, ci, ci, ci, ci, vp]
        L.mgn_abn_maxpool_bwd.argtypes = [vp] * 10 + [cf, cf, ci, cf, ci, ci, ci, ci, vp]
        L.mgn_depth_metrics_workspace_bytes.argtypes = [ci, ci, ctypes.POINTER(sz)]
        L.mgn_depth_metrics.argtypes = [vp, vp, ci, ci, cf, cf, ci, ci, ci, ci, ci, vp, vp, sz, vp]
        L.mgn_depth_post_workspace_bytes.argtypes = [ctypes.POINTER(DepthPostCfg), ctypes.POINTER(sz)]
        L.mgn_depth_post.argtypes = [ctypes.POINTER(DepthPostCfg), vp, vp, vp, vp, vp, vp, sz, vp]
        L.mgn_panoptic_post_workspace_bytes.argtypes = [ctypes.POINTER(PanopticCfg), ctypes.POINTER(sz)]
        L.mgn_panoptic_post.argtypes = [ctypes.POINTER(PanopticCfg), vp, vp, vp, vp, vp, vp, sz, vp]
        L.mgn_panoptic_targets_workspace_bytes.argtypes = [ctypes.POINTER(TargetsCfg), ctypes.POINTER(sz)]
        L.mgn_panoptic_targets.argtypes = [ctypes.POINTER(TargetsCfg)] + [vp] * 15 + [sz, vp]
        L.mgn_geometry_partial_rows.argtypes = [ci, ci, ci, ctypes.POINTER(sz)]
        L.mgn_view_synthesis_fwd.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ci, vp, vp]
        L.mgn_view_synthesis_bwd.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp, vp, vp]
        L.mgn_reconstruct_fwd.argtypes = [vp, vp, vp, ci, ci, ci, vp, vp]
        L.mgn_reconstruct_bwd.argtypes = [vp, vp, vp, vp, ci, ci, ci, vp, vp, vp]
        L.mgn_project_fwd.argtypes = [vp, vp, vp, ci, ci, ci, vp, vp]
        L.mgn_project_bwd.argtypes = [vp, vp, vp, vp, ci, ci, ci, vp, vp, vp]
        L.mgn_plan_begin.argtypes = []
        L.mgn_plan_recorded.argtypes = []
        L.mgn_plan_current.argtypes = []
        L.mgn_plan_abort.argtypes = []
        L.mgn_plan_end.argtypes = [ctypes.POINTER(vp)]
        L.mgn_plan_node_count.argtypes = [vp]
        L.mgn_plan_node_info.argtypes = [vp, ci, ctypes.POINTER(PlanNodeInfo)]
        L.mgn_plan_node_args.argtypes = [vp, ci, ci, ctypes.POINTER(ci), ctypes.POINTER(ci), ctypes.POINTER(ci)]
        L.mgn_plan_compile.argtypes = [vp, ci, ctypes.POINTER(ci), ctypes.POINTER(ci), ctypes.POINTER(vp), ci, ci]
        L.mgn_plan_run.argtypes = [vp, ci, ci]
        L.mgn_plan_set_stream.argtypes = [vp, ci, vp]
        L.mgn_plan_prof_elapsed.argtypes = [vp, ci, ctypes.POINTER(cf)]
        L.mgn_plan_free.argtypes = [vp]
        L.mgn_plan_trace.argtypes = [vp, ci]
        L.mgn_plan_trace_read.argtypes = [vp, ci, ci, ctypes.POINTER(cf), ctypes.POINTER(cf), ctypes.POINTER(cf)]
        L.mgn_plan_set_skip.argtypes = [vp, ci, ci]
        L.mgn_plan_node_ro.argtypes = [vp, ci, ci, ctypes.POINTER(ctypes.c_ulonglong)]
        L.mgn_plan_node_ro_family.argtypes = [vp, ci, ci, ctypes.POINTER(ci)]
        L.mgn_plan_set_jitter.argtypes = [vp, ctypes.c_ulonglong, ci, ci]
        L.mgn_plan_probe.argtypes = [vp, ci, vp, sz, vp]
        L.mgn_abn_apply_pool.argtypes = [vp, vp, vp, vp, ci, cf, ci, cl, ci, cf, vp, vp, sz, vp]
        L.mgn_att_abn_bwd_stats.argtypes = [vp, vp, vp, vp, cf, ci, cf, ci, cl, ci, vp, vp, sz, vp]
        L.mgn_att_abn_bwd_sums.argtypes = [vp, vp, vp, ci, ci, ci, vp, vp, vp, vp]
        L.mgn_att_abn_bwd_apply.argtypes = [vp, vp, vp, vp, vp, ci, vp, vp, vp, vp, cf, cf, ci, cf, ci, cl, ci, vp]
        for n in SYMBOLS[4:]:
            getattr(L, n).restype = ci
        L.mgn_p2p_mailbox_bytes.restype = sz
        L.mgn_plan_current.restype = vp
        for n in F16_TWINS:
            getattr(L, n + "_f16").restype = ci
            getattr(L, n + "_f16").argtypes = getattr(L, n).argtypes
        _lib = L
    return _lib


_NP_OK = (torch.float32, torch.float64, torch.int32, torch.int64, torch.uint8, torch.int16, torch.bool)


class PinnedStager:
    """Host -> device copies of small per-step tensors without stalling the host: a copy from pageable memory is
    stream-ordered AND blocks the host until every kernel queued before it has run.  Staged through a ring of pinned
    buffers; each buffer carries an event recorded after its copy was queued, and is only rewritten once that event has
    completed (the host runs ahead of the GPU, so without it a buffer could be overwritten before its copy executes --
    the wait only ever blocks when the host is a full ring ahead, which is the back-pressure one wants anyway)."""

    def __init__(self, depth=3):
        self.depth, self.rings = depth, {}

    def stage(self, src, device, slot=None, data=False):
        """`slot`: one ring per call site (two sites staging equal shapes in the same step must not share buffers).
        `data`: the tensor carries per-batch VALUES (e.g. the camera matrices of the dataset mapper), not addresses of the step's own
        buffers -- a step that is being recorded for replay must not freeze them."""
        if PLAN_RECORDER[0] is not None and data:
            from .engine.plan import PlanUnsupported
            raise PlanUnsupported(f"a per-batch host tensor ('{slot}') is uploaded inside the step: a replay would reuse the recorded batch's values; "
                                  "hand the step device tensors (Trainer.run_step_planned keeps device copies of host entries and refills them)")
        if PLAN_RECORDER[0] is not None:
            # a step being recorded for replay: the table's content is the same in every replay (addresses of the recorded step), so it is
            # uploaded ONCE, now, into a device tensor the plan keeps alive -- the launch goes to the library directly, past the recorder,
            # and is therefore not part of the replayed schedule
            rec = PLAN_RECORDER[0]
            buf = torch.empty(src.shape, dtype=src.dtype).pin_memory()
            buf.copy_(src)
            out = rec.static_table(src.shape, src.dtype)   # (the plan's own arena: never a block the step's temporaries pass through)
            check(rec.lib.mgn_copy_from_host(out.data_ptr(), buf.data_ptr(), out.numel() * out.element_size(), _stream()), "mgn_copy_from_host")
            rec.keep.append((buf, out))
            return out
        key = (slot, tuple(src.shape), src.dtype, str(device))
        ring = self.rings.get(key)
        if ring is None:
            ring = self.rings[key] = dict(bufs=[torch.empty(src.shape, dtype=src.dtype).pin_memory() for _ in range(self.depth)],
                                          evs=[None] * self.depth, turn=0)
        i = ring["turn"]
        ring["turn"] = (i + 1) % self.depth
        if ring["evs"][i] is not None:
            ring["evs"][i].synchronize()
        # plain memcpy through numpy views: a torch CPU copy_ of more than 32 K elements goes through the intra-op thread pool,
        # whose workers then spin-wait on every core for the next ~200 ms and starve this (launch-issuing) thread -- measured: the
        # single 60 K-float optimizer table made the whole step 10-15 ms slower and erratic
        if src.dtype in _NP_OK and src.is_contiguous() and not src.requires_grad:
            views = ring.get("np")
            if views is None:
                views = ring["np"] = [b.numpy() for b in ring["bufs"]]
            np.copyto(views[i], src.numpy())
        else:
            ring["bufs"][i].copy_(src)
        nb = src.numel() * src.element_size()
        if nb % 4 == 0 and nb > 0:
            # (a kernel that reads the pinned buffer instead of an async memcpy: see mgn_copy_from_host)
            out = torch.empty(src.shape, dtype=src.dtype, device=device)
            check(lib().mgn_copy_from_host(out.data_ptr(), ring["bufs"][i].data_ptr(), nb, _stream()), "mgn_copy_from_host")
        else:
            out = ring["bufs"][i].to(device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        ring["evs"][i] = ev
        return out


def _stage_into(self, dst, src, slot=None):
    """like stage(), straight into the existing device tensor `dst` (contiguous, same dtype and element count)"""
    assert dst.is_cuda and dst.is_contiguous() and dst.dtype == src.dtype and dst.numel() == src.numel() and (src.numel() * src.element_size()) % 4 == 0
    key = (slot, tuple(src.shape), src.dtype, "into")
    ring = self.rings.get(key)
    if ring is None:
        ring = self.rings[key] = dict(bufs=[torch.empty(src.shape, dtype=src.dtype).pin_memory() for _ in range(self.depth)],
                                      evs=[None] * self.depth, turn=0)
        ring["np"] = [b.numpy() for b in ring["bufs"]] if src.dtype in _NP_OK else None
    i = ring["turn"]
    ring["turn"] = (i + 1) % self.depth
    if ring["evs"][i] is not None:
        ring["evs"][i].synchronize()
    if ring["np"] is not None and src.is_contiguous() and not src.requires_grad:
        np.copyto(ring["np"][i], src.numpy())
    else:
        ring["bufs"][i].copy_(src)
    check(lib().mgn_copy_from_host(dst.data_ptr(), ring["bufs"][i].data_ptr(), src.numel() * src.element_size(), _stream()), "mgn_copy_from_host")
    ev = torch.cuda.Event()
    ev.record()
    ring["evs"][i] = ev
    return dst


PinnedStager.stage_into = _stage_into


def check(rc, what):
    if rc != 0:
        err = _ERR.get(rc, str(rc))
        if rc == -95:
            raise NotImplementedError(f"{what}: {err}")
        raise RuntimeError(f"{what}: {err}")


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * MGN_MAX_SCALES)()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


def _dev_f32(t, name):
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError(f"{name}: expected a contiguous float32 tensor on the GPU, got {t.dtype} {t.device} "
                         f"contiguous={t.is_contiguous()}")
    return t


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """hipStream_t of torch's CURRENT stream of the current device (honours torch.cuda.stream(...) contexts).  The raw
    getter avoids building a torch.cuda.Stream object on each of the ~900 launches of a step."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_PAD = {"zeros": 0, "border": 1, "reflection": 2}
_RED = {"min": 0, "mean": 1}


def make_reproj_cfg(B, H, W, n_scales, ssim_w=0.85, photo_w=1.0, smooth_w=0.001, automask=True, reduce_op="min",
                    padding_mode="zeros", rows_per_wave=0):
    return ReprojCfg(B, H, W, n_scales, ssim_w, photo_w, smooth_w, int(bool(automask)), _RED[reduce_op],
                     _PAD[padding_mode], rows_per_wave, 0, None, None)


def reproj_workspace_bytes(cfg):
    n = ctypes.c_size_t(0)
    check(lib().mgn_reproj_workspace_bytes(ctypes.byref(cfg), ctypes.byref(n)), "mgn_reproj_workspace_bytes")
    return n.value


def reproj_loss_fwd(cfg, inv, img, prev, nxt, mask, cam, pose, want_grad=True, want_minmap=False):
    """-> dict(losses[2], d_pose[B,2,6], g_inv[list], workspace, minmap)   (all device tensors, stream-ordered)"""
    dev = img.device
    inv = [_dev_f32(t, f"inv_depth[{i}]") for i, t in enumerate(inv)]
    for n_, t in (("camera_matrix", cam), ("pose", pose)):
        _dev_f32(t, n_)
    # frame layouts (mgn_reproj_cfg.frame_layout): [B,3,H,W] fp32 planes like the reference's tensors; or both context frames
    # pixel-interleaved fp32 RGBx (a [B,4,H,W] channels_last tensor = [B,H,W,4] in memory, 4th channel ignored); or ALL THREE frames
    # as uint8 RGBX ([B,4,H,W] uint8 channels_last, what u8_frames_to_rgbx makes of the uint8 frames the step receives)
    nhwc4 = lambda t: t.dim() == 4 and t.shape[1] == 4 and t.is_cuda and t.is_contiguous(memory_format=torch.channels_last)
    if all(nhwc4(t) and t.dtype == torch.uint8 for t in (img, prev, nxt)):
        cfg.frame_layout = 2
    else:
        _dev_f32(img, "img")
        ilv = all(nhwc4(t) for t in (prev, nxt))
        for n_, t in (("prev", prev), ("next", nxt)):
            if not (t.is_cuda and t.dtype == torch.float32 and (ilv or (t.shape[1] == 3 and t.is_contiguous()))):
                raise ValueError(f"{n_}: expected a float32 GPU tensor, [B,3,H,W] contiguous or (both context frames) [B,4,H,W] channels_last; "
                                 "or img, prev and next all uint8 [B,4,H,W] channels_last")
        cfg.frame_layout = int(ilv)
    if mask is not None:
        if mask.dtype not in (torch.bool, torch.uint8) or not mask.is_contiguous() or not mask.is_cuda:
            raise ValueError("reprojection_mask: expected a contiguous bool/uint8 GPU tensor")
    cam_ld = cam.shape[-1]
    cam_stride = cam.shape[-1] * cam.shape[-2]
    ws = torch.empty(reproj_workspace_bytes(cfg), dtype=torch.uint8, device=dev)
    losses = torch.empty(2, dtype=torch.float32, device=dev)
    d_pose = torch.empty((cfg.B, 2, 6), dtype=torch.float32, device=dev) if want_grad else None
    g_inv = [torch.empty_like(t) for t in inv] if want_grad else None
    minmap = torch.zeros((cfg.n_scales, cfg.B, 1, cfg.H, cfg.W), dtype=torch.float32, device=dev) if want_minmap else None
    rc = lib().mgn_reproj_loss_fwd(
        ctypes.byref(cfg), _ptr_array(inv), img.data_ptr(), prev.data_ptr(), nxt.data_ptr(),
        None if mask is None else mask.data_ptr(), cam.data_ptr(), cam_stride, cam_ld, pose.data_ptr(), int(want_grad),
        losses.data_ptr(), None if d_pose is None else d_pose.data_ptr(), None if g_inv is None else _ptr_array(g_inv),
        None if minmap is None else minmap.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
    check(rc, "mgn_reproj_loss_fwd")
    return {"losses": losses, "d_pose": d_pose, "g_inv": g_inv, "workspace": ws, "minmap": minmap}


def reproj_loss_bwd(cfg, inv, img, mask, grad_losses, fwd):
    """Finishes the backward IN PLACE on fwd['g_inv']; returns (d_inv list, d_pose)."""
    _dev_f32(grad_losses, "grad_losses")
    d_pose_out = torch.empty_like(fwd["d_pose"])
    rc = lib().mgn_reproj_loss_bwd(
        ctypes.byref(cfg), _ptr_array(inv), img.data_ptr(), None if mask is None else mask.data_ptr(),
        grad_losses.data_ptr(), fwd["d_pose"].data_ptr(), _ptr_array(fwd["g_inv"]), d_pose_out.data_ptr(),
        fwd["workspace"].data_ptr(), fwd["workspace"].numel(), _stream())
    check(rc, "mgn_reproj_loss_bwd")
    return fwd["g_inv"], d_pose_out


# ---------------------------------------------------------------------------------------------------------------
# activated batch norm
# ---------------------------------------------------------------------------------------------------------------
_IABN_WS = {}


def _iabn_ws(device):
    """partials workspace of the single-launch reductions: one per (device, stream) -- launches on the same stream are
    ordered, launches on different streams (two encoders side by side, multi-scale eval) must not share partials."""
    key = (device, _stream().value)
    ws = _IABN_WS.get(key)
    if ws is None:
        ws = _IABN_WS[key] = torch.empty(2 * 1024 * 1024, dtype=torch.float32, device=device)  # 2*C*1024 floats, C<=1024
    return ws


def _act_dtype(t):
    if t.dtype == torch.float32:
        return 0
    if t.dtype in H16:   # 1 = "the 16-bit format of the entry point": bf16, or fp16 in the _f16 twins
        return 1
    raise ValueError(f"activations must be float32, bfloat16 or float16, got {t.dtype}")


def iabn_stats(x2d_like, M, C):
    """x: channels-last activation storage -> stats[3,C] = (count, mean, M2) of this rank."""
    stats = torch.empty((3, C), dtype=torch.float32, device=x2d_like.device)
    ws = _iabn_ws(x2d_like.device)
    check(_fn("mgn_iabn_stats", x2d_like)(x2d_like.data_ptr(), _act_dtype(x2d_like), M, C, stats.data_ptr(), ws.data_ptr(),
                               ws.numel() * 4, _stream()), "mgn_iabn_stats")
    return stats


def iabn_train_coeffs(x, M, C, weight, bias, eps, momentum, running_mean, running_var):
    """single-rank training forward: batch statistics -> coef[4,C] = (scale, offset, mean, rstd) in one launch"""
    out = torch.empty((4, C), dtype=torch.float32, device=x.device)
    ws = _iabn_ws(x.device)
    check(_fn("mgn_iabn_train_coeffs", x)(x.data_ptr(), _act_dtype(x), M, C, weight.data_ptr(), bias.data_ptr(), eps, momentum,
                                      None if running_mean is None else running_mean.data_ptr(),
                                      None if running_var is None else running_var.data_ptr(), out.data_ptr(),
                                      ws.data_ptr(), ws.numel() * 4, _stream()), "mgn_iabn_train_coeffs")
    return out


def iabn_combine(gathered, weight, bias, eps, momentum, running_mean, running_var):
    R, _, C = gathered.shape
    out = torch.empty((4, C), dtype=torch.float32, device=gathered.device)  # scale, offset, mean, rstd
    check(lib().mgn_iabn_combine(gathered.data_ptr(), R, C, weight.data_ptr(), bias.data_ptr(), eps, momentum,
                                 None if running_mean is None else running_mean.data_ptr(),
                                 None if running_var is None else running_var.data_ptr(), out[0].data_ptr(),
                                 out[1].data_ptr(), out[2].data_ptr(), _stream()), "mgn_iabn_combine")
    return out


def iabn_eval_coeffs(weight, bias, running_mean, running_var, eps):
    C = weight.numel()
    out = torch.empty((2, C), dtype=torch.float32, device=weight.device)
    check(lib().mgn_iabn_eval_coeffs(C, weight.data_ptr(), bias.data_ptr(), running_mean.data_ptr(),
                                     running_var.data_ptr(), eps, out[0].data_ptr(), out[1].data_ptr(), _stream()),
          "mgn_iabn_eval_coeffs")
    return out


def iabn_apply(x, y, M, C, scale, offset, activation, slope):
    check(_fn("mgn_iabn_apply", x)(x.data_ptr(), y.data_ptr(), _act_dtype(x), M, C, scale.data_ptr(), offset.data_ptr(),
                               activation, slope, _stream()), "mgn_iabn_apply")


def iabn_bwd_reduce(y, dy, M, C, weight, bias, eps, activation, slope):
    out = torch.empty((4, C), dtype=torch.float32, device=y.device)  # sums[2], d_weight, d_bias
    ws = _iabn_ws(y.device)
    check(_fn("mgn_iabn_bwd_reduce", y)(y.data_ptr(), dy.data_ptr(), _act_dtype(y), M, C, weight.data_ptr(), bias.data_ptr(),
                                    eps, activation, slope, out.data_ptr(), out[2].data_ptr(), ws.data_ptr(), ws.numel() * 4,
                                    _stream()), "mgn_iabn_bwd_reduce")
    return out[:2], out[2], out[3]


def iabn_bwd_reduce_x(x, dy, M, C, weight, bias, coef, eps, activation, slope):
    """like iabn_bwd_reduce, from the norm's input x (z = coef[0] * x + coef[1] recomputed)"""
    out = torch.empty((4, C), dtype=torch.float32, device=x.device)
    ws = _iabn_ws(x.device)
    check(_fn("mgn_iabn_bwd_reduce_x", x)(x.data_ptr(), dy.data_ptr(), _act_dtype(x), M, C, weight.data_ptr(), bias.data_ptr(),
                                      coef[0].data_ptr(), coef[1].data_ptr(), eps, activation, slope, out.data_ptr(),
                                      out[2].data_ptr(), ws.data_ptr(), ws.numel() * 4, _stream()), "mgn_iabn_bwd_reduce_x")
    return out[:2], out[2], out[3]


def iabn_bwd_reduce_x_relu(x, g, yrelu, M, C, weight, bias, coef, eps, relu_bits=None):
    """block tail: dm = g * (yrelu > 0) written AND reduced in one pass (mgn_iabn_bwd_reduce_x_relu) -> (dm, sums[2,C], d_weight, d_bias);
    with `relu_bits` (abn_add_relu_fwd(..., want_bits=True)) the mask is read from there instead of yrelu"""
    out = torch.empty((4, C), dtype=torch.float32, device=x.device)
    dm = _cl_like(x)
    ws = _iabn_ws(x.device)
    check(_fn("mgn_iabn_bwd_reduce_x_relu", x)(x.data_ptr(), g.data_ptr(), None if yrelu is None else yrelu.data_ptr(),
                                           None if relu_bits is None else relu_bits.data_ptr(), dm.data_ptr(), M, C, weight.data_ptr(),
                                           bias.data_ptr(), coef[0].data_ptr(), coef[1].data_ptr(), eps, out.data_ptr(), out[2].data_ptr(),
                                           ws.data_ptr(), ws.numel() * 4, _stream()), "mgn_iabn_bwd_reduce_x_relu")
    return dm, out[:2], out[2], out[3]


def iabn_bwd_apply_x(x, dy, dx, M, C, weight, bias, coef, sums, total_count, eps, activation, slope):
    check(_fn("mgn_iabn_bwd_apply_x", x)(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), _act_dtype(x), M, C, weight.data_ptr(),
                                     bias.data_ptr(), coef[0].data_ptr(), coef[1].data_ptr(), coef[2:].data_ptr(), sums.data_ptr(),
                                     float(total_count), eps, activation, slope, _stream()), "mgn_iabn_bwd_apply_x")


def abn_add_relu_fwd(x, coef, shortcut, want_bits=False):
    """y = relu(norm(x) + shortcut); want_bits: also the ReLU mask, one byte per 8 outputs -> (y, bits)"""
    N, C, H, W = x.shape
    y = torch.empty_like(x)
    bits = torch.empty(N * H * W * C // 8, dtype=torch.uint8, device=x.device) if want_bits else None
    check(_fn("mgn_abn_add_relu_fwd", x)(x.data_ptr(), coef[0].data_ptr(), coef[1].data_ptr(), shortcut.data_ptr(), y.data_ptr(),
                                     None if bits is None else bits.data_ptr(), N * H * W, C, _stream()), "mgn_abn_add_relu_fwd")
    return (y, bits) if want_bits else y


def iabn_bwd_apply(y, dy, dx, M, C, weight, bias, saved, sums, total_count, eps, activation, slope):
    check(_fn("mgn_iabn_bwd_apply", y)(y.data_ptr(), dy.data_ptr(), dx.data_ptr(), _act_dtype(y), M, C, weight.data_ptr(),
                                   bias.data_ptr(), saved.data_ptr(), sums.data_ptr(), float(total_count), eps,
                                   activation, slope, _stream()), "mgn_iabn_bwd_apply")


# ---------------------------------------------------------------------------------------------------------------
# clip + Adam over flat buckets
# ---------------------------------------------------------------------------------------------------------------
def optim_chunk():
    return lib().mgn_optim_chunk()


def sqnorm(g, partials, offset):
    """block partials of sum g^2 into partials[offset:]; returns the number of partials written"""
    n = ctypes.c_int(0)
    check(lib().mgn_sqnorm(g.data_ptr(), g.numel(), partials.data_ptr() + 4 * offset, partials.numel() - offset,
                           ctypes.byref(n), _stream()), "mgn_sqnorm")
    return n.value


def clip_coef(partials, n_partials, max_norm, grad_scale, out):
    check(lib().mgn_clip_coef(partials.data_ptr(), n_partials, max_norm, grad_scale, out.data_ptr(), _stream()), "mgn_clip_coef")


def adam_step(p, g, m, v, chunk_lr, chunk_wd, beta1, beta2, eps, step, coef, grad_scale):
    check(lib().mgn_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), chunk_lr.data_ptr(),
                              chunk_wd.data_ptr(), beta1, beta2, eps, step, coef.data_ptr(), grad_scale, _stream()),
          "mgn_adam_step")


def clip_coef_scaled(partials, n, max_norm, grad_scale, beta1, beta2, growth_interval, scaler_state, hyper, coef):
    check(lib().mgn_clip_coef_scaled(partials.data_ptr(), n, max_norm, grad_scale, beta1, beta2, int(growth_interval),
                                     scaler_state.data_ptr(), hyper.data_ptr(), coef.data_ptr(), _stream()), "mgn_clip_coef_scaled")


def adam_step_dev(p, g, m, v, chunk_lr, chunk_wd, beta1, beta2, eps, hyper, coef, grad_scale):
    """Adam with the bias corrections in device memory (`hyper` = [1/(1-b1^t), 1/sqrt(1-b2^t)]): graph-capturable"""
    check(lib().mgn_adam_step_dev(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), chunk_lr.data_ptr(),
                                  chunk_wd.data_ptr(), beta1, beta2, eps, hyper.data_ptr(), coef.data_ptr(), grad_scale, _stream()),
          "mgn_adam_step_dev")


OPTIM_KIND = {"ADAM": 0, "ADAMW": 1, "SGD": 2}


def optim_step_dev(kind, p, g, m, v, chunk_lr, chunk_wd, beta1, beta2, eps, nesterov, hyper, coef, grad_scale):
    """Adam | AdamW | SGD (momentum) on one flat bucket (mgn_optim_step_dev); v may be None for SGD"""
    check(lib().mgn_optim_step_dev(OPTIM_KIND[kind], p.data_ptr(), g.data_ptr(), m.data_ptr(), None if v is None else v.data_ptr(), p.numel(),
                                   chunk_lr.data_ptr(), chunk_wd.data_ptr(), beta1, beta2, eps, int(bool(nesterov)), hyper.data_ptr(),
                                   coef.data_ptr(), grad_scale, _stream()), "mgn_optim_step_dev")


# ---------------------------------------------------------------------------------------------------------------
# convolution (implicit GEMM, bf16 MFMA).  Tensors are logical NCHW in channels_last memory format.
# ---------------------------------------------------------------------------------------------------------------
def conv_supported(x, weight):
    """Cin % 32 == 0, or a channel-padded stem input (Cin 4/8/16 holding the weight's 3/9 real channels); 8-byte pixels (4 channels)
    exist for the 7x7 dense-row stem only (csrc/conv_stem.hip; weight_layout mode 2 with Cp = 4 refuses other kernel sizes)"""
    if not (x.is_cuda and x.dtype in H16):
        return False
    if weight.shape[1] % 32 == 0:
        return True
    if x.shape[1] == 4:
        return weight.shape[1] <= 4 and tuple(weight.shape[2:]) == (7, 7)
    return x.shape[1] in (8, 16) and weight.shape[1] <= x.shape[1]


def stem_input_channels(N, H, W, real=3):
    """channel padding of the 7x7 / stride-2 stem's input: 4 where the dense-row kernel takes the shape (3 real channels, even width:
    csrc/conv_stem.hip CP = 4), else 8 (16 for the 9-channel pose stem)"""
    if real > 3:
        return 16 if real > 8 else 8
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    return 4 if lib().mgn_conv_stem7_blocks(N, H, W, 4, OH, OW, 64) > 0 else 8


def u8_frames_to_f32(frames, divisor):
    """list of equal-shape uint8 CUDA tensors -> stacked fp32 batch `frame / divisor` (one launch), or None if unsupported"""
    f0 = frames[0]
    n = f0.numel()
    if (len(frames) > 16 or n % 16 or any(t.dtype != torch.uint8 or not t.is_cuda or t.shape != f0.shape or not t.is_contiguous()
                                          or t.data_ptr() % 16 for t in frames)):
        return None
    out = torch.empty((len(frames),) + tuple(f0.shape), dtype=torch.float32, device=f0.device)
    ptrs = (ctypes.c_void_p * len(frames))(*[t.data_ptr() for t in frames])
    check(lib().mgn_u8_frames_to_f32(ptrs, len(frames), n, float(divisor), out.data_ptr(), _stream()), "mgn_u8_frames_to_f32")
    return out


def u8_frames_to_f32_rgbx(frames, divisor):
    """list of [3,H,W] uint8 CUDA frames -> [B,4,H,W] fp32 channels_last batch ([B,H,W,4] in memory, `frame / divisor`, 4th channel 0):
    the pixel-interleaved context-frame layout of the reprojection loss; None if unsupported"""
    f0 = frames[0]
    if (len(frames) > 16 or f0.dim() != 3 or f0.shape[0] != 3 or (f0.shape[1] * f0.shape[2]) % 4
            or any(t.dtype != torch.uint8 or not t.is_cuda or t.shape != f0.shape or not t.is_contiguous() or t.data_ptr() % 4 for t in frames)):
        return None
    H, W = f0.shape[1:]
    out = torch.empty((len(frames), 4, H, W), dtype=torch.float32, device=f0.device, memory_format=torch.channels_last)
    ptrs = (ctypes.c_void_p * len(frames))(*[t.data_ptr() for t in frames])
    check(lib().mgn_u8_frames_to_f32_nhwc4(ptrs, len(frames), H * W, float(divisor), out.data_ptr(), _stream()), "mgn_u8_frames_to_f32_nhwc4")
    return out


def u8_frames_to_rgbx(frames):
    """list of n <= 48 [3,H,W] uint8 CUDA frames -> [n,4,H,W] uint8 channels_last batch ([n,H,W,4] in memory: R,G,B,0), one launch; None if
    unsupported.  The frame layout MGN_FRAMES_RGBX_U8 of the reprojection loss."""
    f0 = frames[0]
    if (len(frames) > 48 or f0.dim() != 3 or f0.shape[0] != 3 or (f0.shape[1] * f0.shape[2]) % 4
            or any(t.dtype != torch.uint8 or not t.is_cuda or t.shape != f0.shape or not t.is_contiguous() or t.data_ptr() % 4 for t in frames)):
        return None
    H, W = f0.shape[1:]
    out = torch.empty((len(frames), 4, H, W), dtype=torch.uint8, device=f0.device, memory_format=torch.channels_last)
    ptrs = (ctypes.c_void_p * len(frames))(*[t.data_ptr() for t in frames])
    check(lib().mgn_u8_frames_to_rgbx(ptrs, len(frames), H * W, out.data_ptr(), _stream()), "mgn_u8_frames_to_rgbx")
    return out


_HEAD_ACT = {"none": 0, "sigmoid": 1, "sigmoid2": 2}


def head_act_fwd(xp, C, kind):
    """xp: padded predictor output [B,P,h,w] 16-bit channels_last -> fp32 [B,C,h,w] = act(xp[:, :C])   (mgn_head_act_fwd)"""
    B, P, h, w = xp.shape
    assert xp.is_cuda and xp.dtype in H16 and xp.is_contiguous(memory_format=torch.channels_last) and C <= P
    y = torch.empty((B, C, h, w), dtype=torch.float32, device=xp.device)
    check(lib().mgn_head_act_fwd(xp.data_ptr(), B, h, w, P, C, _HEAD_ACT[kind], int(xp.dtype == torch.float16), y.data_ptr(), _stream()), "mgn_head_act_fwd")
    return y


def head_act_bwd(g, y, shape, C, kind, dtype, gscale=1.0, g_strides=None):
    """g: fp32 gradient wrt act(x)[:, :C] ([B,C,h,w]-indexable through element strides (sb, sc, sp)) -> gradient wrt the padded predictor
    output, [B,P,h,w] `dtype` channels_last with zero padding channels   (mgn_head_act_bwd)"""
    B, P, h, w = shape
    if g_strides is None:
        assert g.shape == (B, C, h, w) and g.stride(3) * w == g.stride(2)
        g_strides = (g.stride(0), g.stride(1), g.stride(3))
    dx = torch.empty((B, P, h, w), dtype=dtype, device=g.device, memory_format=torch.channels_last)
    check(lib().mgn_head_act_bwd(_dev_f32_any(g).data_ptr(), g_strides[0], g_strides[1], g_strides[2], None if y is None else y.data_ptr(), B, h, w, P, C,
                                 _HEAD_ACT[kind], int(dtype == torch.float16), float(gscale), dx.data_ptr(), _stream()), "mgn_head_act_bwd")
    return dx


def _dev_f32_any(t):
    if not (t.is_cuda and t.dtype == torch.float32):
        raise ValueError(f"expected a float32 GPU tensor, got {t.dtype} {t.device}")
    return t


def uncertainty_fwd(raws, log_vars, tau_mask):
    """raws: list of n fp32 device scalars -> (weighted [n], uncertainty [n])   (mgn_uncertainty_fwd)"""
    n = len(raws)
    out = torch.empty((2, n), dtype=torch.float32, device=log_vars.device)
    ptrs = (ctypes.c_void_p * n)(*[_dev_f32(t, "raw loss").data_ptr() for t in raws])
    check(lib().mgn_uncertainty_fwd(ptrs, n, _dev_f32(log_vars, "log_vars").data_ptr(), tau_mask, out[0].data_ptr(), out[1].data_ptr(), _stream()),
          "mgn_uncertainty_fwd")
    return out[0], out[1]


def uncertainty_bwd(raws, grads, log_vars, tau_mask):
    """-> (d_raw [n], d_log_vars [len(log_vars)])"""
    n = len(raws)
    d_raw = torch.empty(n, dtype=torch.float32, device=log_vars.device)
    d_lv = torch.empty(log_vars.numel(), dtype=torch.float32, device=log_vars.device)
    rp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in raws])
    gp = (ctypes.c_void_p * n)(*[None if g is None else _dev_f32(g, "grad").data_ptr() for g in grads])
    check(lib().mgn_uncertainty_bwd(rp, gp, n, log_vars.numel(), log_vars.data_ptr(), tau_mask, d_raw.data_ptr(), d_lv.data_ptr(), _stream()),
          "mgn_uncertainty_bwd")
    return d_raw, d_lv


_MSC_MODE = {"softmax": 0, "plain": 1, "offset": 2, "inv2depth": 3}
_MSC_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}


def msc_input(norm, h, w, flip, dtype):
    """[N,3,H,W] fp32 normalised frames -> bilinear(align_corners) rescale to (h, w) (+ horizontal flip) as the network input
    [N,8,h,w] `dtype` channels_last (channels 3..7 zero): one launch instead of interpolate + flip + pad + cast + layout change"""
    N, C, H, W = norm.shape
    assert C == 3 and norm.dtype == torch.float32 and norm.is_cuda and norm.is_contiguous() and (dtype in H16 or dtype == torch.float32)
    # 16-bit trunks: 8 channels (3..7 zero, the packed-tap stem's layout); fp32 trunk: the 3 real channels, channels_last
    out = torch.empty((N, 3 if dtype == torch.float32 else 8, h, w), dtype=dtype, device=norm.device, memory_format=torch.channels_last)
    code = 2 if dtype == torch.float32 else int(dtype == torch.float16)
    check(lib().mgn_msc_input(norm.data_ptr(), N, H, W, h, w, int(bool(flip)), code, out.data_ptr(), _stream()), "mgn_msc_input")
    return out


def msc_accumulate(acc, lr, mode, flip, first, stride=1.0, scale=1.0, divide=0.0):
    """acc[N,C,H,W] fp32 (+)= f(bilinear upsample of lr[N,C,h,w]) for one pass of multi-scale + flip inference (mgn_msc_accumulate)"""
    N, C, h, w = lr.shape
    assert acc.is_cuda and acc.dtype == torch.float32 and acc.is_contiguous() and acc.shape[:2] == (N, C) and lr.dtype in _MSC_DT and C <= 32
    H, W = acc.shape[2:]
    check(lib().mgn_msc_accumulate(lr.data_ptr(), _MSC_DT[lr.dtype], lr.stride(0), lr.stride(1), lr.stride(2), lr.stride(3), N, C, h, w, H, W,
                                   _MSC_MODE[mode], int(bool(flip)), int(bool(first)), float(stride), float(scale), float(divide),
                                   acc.data_ptr(), _stream()), "mgn_msc_accumulate")
    return acc


def prep_input(frames_u8, mean3, std3, Cp, dtype=torch.bfloat16):
    """frames: list of [B,3,H,W] uint8 CUDA tensors -> [B,Cp,H,W] bf16 / fp16 channels_last (normalised, zero-padded channels; Cp = 4 | 8 | 16)"""
    B, _, H, W = frames_u8[0].shape
    frames_u8 = [f.contiguous() for f in frames_u8]
    out = torch.empty((B, Cp, H, W), dtype=dtype, device=frames_u8[0].device, memory_format=torch.channels_last)
    ptrs = (ctypes.c_void_p * 3)(*[f.data_ptr() for f in frames_u8])
    m = (ctypes.c_float * 3)(*[float(v) for v in mean3])
    sd = (ctypes.c_float * 3)(*[float(v) for v in std3])
    check(_fn("mgn_prep_input", out)(ptrs, len(frames_u8), B, H, W, m, sd, out.data_ptr(), Cp, _stream()), "mgn_prep_input")
    return out


def panoptic_post(cfg, sem_seg, center, offsets):
    """mgn_panoptic_post: sem_seg int64 [H,W], center f32 [H,W], offsets f32 [2,H,W] (CUDA, contiguous) ->
    (panoptic int64 [H,W], info int32[2] on the device)."""
    H, W = cfg.H, cfg.W
    assert sem_seg.is_cuda and sem_seg.dtype == torch.int64 and sem_seg.is_contiguous() and tuple(sem_seg.shape) == (H, W)
    assert center.dtype == torch.float32 and center.is_contiguous() and tuple(center.shape) == (H, W)
    assert offsets.dtype == torch.float32 and offsets.is_contiguous() and tuple(offsets.shape) == (2, H, W)
    nbytes = ctypes.c_size_t()
    check(lib().mgn_panoptic_post_workspace_bytes(ctypes.byref(cfg), ctypes.byref(nbytes)), "mgn_panoptic_post_workspace_bytes")
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=sem_seg.device)
    pan = torch.empty((H, W), dtype=torch.int64, device=sem_seg.device)
    info = torch.empty(2, dtype=torch.int32, device=sem_seg.device)
    check(lib().mgn_panoptic_post(ctypes.byref(cfg), sem_seg.data_ptr(), center.data_ptr(), offsets.data_ptr(), pan.data_ptr(),
                                  info.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "mgn_panoptic_post")
    return pan, info


INSTANCE_MAX = 4096


class InstanceCfg(ctypes.Structure):
    _fields_ = [("H", ctypes.c_int), ("W", ctypes.c_int), ("C", ctypes.c_int), ("label_divisor", ctypes.c_int),
                ("thing_mask", ctypes.c_uint64)]


def instance_post(sem_logits, center, panoptic, thing_ids, label_divisor, want_masks=True):
    """mgn_instance_post (+ mgn_instance_masks): sem_logits f32 [C,H,W], center f32 [H,W], panoptic int64 [H,W] (CUDA) ->
    (labels int64 [n], classes int64 [n], scores f32 [n], boxes f32 [n,4], masks bool [n,H,W] or None).  One host read-back of n
    (the reference copies the whole panoptic image to the host for np.unique)."""
    C, H, W = sem_logits.shape
    assert sem_logits.is_cuda and sem_logits.dtype == torch.float32 and sem_logits.is_contiguous()
    assert center.dtype == torch.float32 and center.is_contiguous() and tuple(center.shape) == (H, W)
    assert panoptic.dtype == torch.int64 and panoptic.is_contiguous() and tuple(panoptic.shape) == (H, W)
    mask = 0
    for t in thing_ids:
        if not 0 <= int(t) < 64:
            raise ValueError(f"thing id {t} outside [0, 64)")
        mask |= 1 << int(t)
    cfg = InstanceCfg(H, W, C, int(label_divisor), mask)
    L = lib()
    L.mgn_instance_post_workspace_bytes.argtypes = [ctypes.POINTER(InstanceCfg), ctypes.POINTER(ctypes.c_size_t)]
    L.mgn_instance_post.argtypes = [ctypes.POINTER(InstanceCfg)] + [ctypes.c_void_p] * 9 + [ctypes.c_size_t, ctypes.c_void_p]
    L.mgn_instance_masks.argtypes = [ctypes.POINTER(InstanceCfg), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    nbytes = ctypes.c_size_t()
    check(L.mgn_instance_post_workspace_bytes(ctypes.byref(cfg), ctypes.byref(nbytes)), "mgn_instance_post_workspace_bytes")
    dev = sem_logits.device
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
    labels = torch.empty(INSTANCE_MAX, dtype=torch.int64, device=dev)
    classes = torch.empty(INSTANCE_MAX, dtype=torch.int32, device=dev)
    scores = torch.empty(INSTANCE_MAX, dtype=torch.float32, device=dev)
    boxes = torch.empty((INSTANCE_MAX, 4), dtype=torch.float32, device=dev)
    info = torch.empty(2, dtype=torch.int32, device=dev)
    check(L.mgn_instance_post(ctypes.byref(cfg), sem_logits.data_ptr(), center.data_ptr(), panoptic.data_ptr(), labels.data_ptr(),
                              classes.data_ptr(), scores.data_ptr(), boxes.data_ptr(), info.data_ptr(), ws.data_ptr(), ws.numel(), _stream()),
          "mgn_instance_post")
    n, overflow = (int(v) for v in info.tolist())
    if overflow:
        raise RuntimeError(f"more than {INSTANCE_MAX} thing segments in one panoptic image")
    masks = None
    if want_masks and n > 0:
        masks = torch.zeros((n, H, W), dtype=torch.uint8, device=dev)
        check(L.mgn_instance_masks(ctypes.byref(cfg), panoptic.data_ptr(), labels.data_ptr(), n, masks.data_ptr(), _stream()), "mgn_instance_masks")
        masks = masks.view(torch.bool)
    return labels[:n], classes[:n].long(), scores[:n], boxes[:n], masks


def pseudo_label_ids(panoptic, label_divisor, id_map):
    """mgn_pseudo_label_ids: panoptic int64 [H,W] (CUDA, train ids) + id_map (256 ints: trainId -> dataset id) -> int16-typed uint16 image
    (torch has no uint16 arithmetic: the tensor is int16 storage holding the uint16 bit patterns; `.cpu().numpy().view(np.uint16)`)."""
    assert panoptic.is_cuda and panoptic.dtype == torch.int64 and panoptic.is_contiguous()
    lut = torch.as_tensor(id_map, dtype=torch.uint8).reshape(256).to(panoptic.device)
    out = torch.empty(panoptic.shape, dtype=torch.int16, device=panoptic.device)
    L = lib()
    L.mgn_pseudo_label_ids.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    check(L.mgn_pseudo_label_ids(panoptic.data_ptr(), panoptic.numel(), int(label_divisor), lut.data_ptr(), out.data_ptr(), _stream()),
          "mgn_pseudo_label_ids")
    return out


def depth_metrics(pred, label, min_depth, max_depth, use_gt_scale, crop):
    """mgn_depth_metrics: pred, label f32 [H,W] (CUDA) -> device f64[9] (abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3, ratio, n)"""
    H, W = label.shape
    assert pred.is_cuda and pred.dtype == torch.float32 and pred.is_contiguous() and tuple(pred.shape) == (H, W)
    assert label.is_cuda and label.dtype == torch.float32 and label.is_contiguous()
    nbytes = ctypes.c_size_t()
    check(lib().mgn_depth_metrics_workspace_bytes(H, W, ctypes.byref(nbytes)), "mgn_depth_metrics_workspace_bytes")
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=pred.device)
    out = torch.empty(9, dtype=torch.float64, device=pred.device)
    check(lib().mgn_depth_metrics(pred.data_ptr(), label.data_ptr(), H, W, float(min_depth), float(max_depth), int(use_gt_scale),
                                  int(crop[0]), int(crop[1]), int(crop[2]), int(crop[3]), out.data_ptr(), ws.data_ptr(),
                                  ws.numel(), _stream()), "mgn_depth_metrics")
    return out


def depth_post(cfg, depth, panoptic=None):
    """mgn_depth_post: depth f32 [H,W] (+ panoptic int64 [H,W]) -> (depth [H,W], xyz [3,H,W] or None, scale f32[1])."""
    H, W = cfg.H, cfg.W
    assert depth.is_cuda and depth.dtype == torch.float32 and depth.is_contiguous() and tuple(depth.shape) == (H, W)
    if panoptic is not None:
        assert panoptic.dtype == torch.int64 and panoptic.is_contiguous() and tuple(panoptic.shape) == (H, W)
    nbytes = ctypes.c_size_t()
    check(lib().mgn_depth_post_workspace_bytes(ctypes.byref(cfg), ctypes.byref(nbytes)), "mgn_depth_post_workspace_bytes")
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=depth.device)
    out = torch.empty_like(depth)
    xyz = torch.empty((3, H, W), dtype=torch.float32, device=depth.device) if cfg.use_dgc_scaling else None
    scale = torch.empty(1, dtype=torch.float32, device=depth.device)
    check(lib().mgn_depth_post(ctypes.byref(cfg), depth.data_ptr(), None if panoptic is None else panoptic.data_ptr(),
                               out.data_ptr(), None if xyz is None else xyz.data_ptr(), scale.data_ptr(), ws.data_ptr(),
                               ws.numel(), _stream()), "mgn_depth_post")
    return out, xyz, scale


def panoptic_targets(cfg, panoptic, seg_ids, seg_attr, seg_count, gauss, want_mask=False, want_points=False):
    """Device-side PanopticDeepLabTargetGenerator (csrc/targets.hip).  cfg: TargetsCfg; panoptic int32 [B,H,W] or uint8
    [B,H,W,3]; seg_* int32 device tables; gauss fp32 device patch.  Returns the dict of batched target maps."""
    B, H, W = cfg.B, cfg.H, cfg.W
    dev = panoptic.device
    assert panoptic.is_cuda and panoptic.is_contiguous()
    assert tuple(panoptic.shape) == ((B, H, W, 3) if cfg.pan_rgb else (B, H, W))
    assert panoptic.dtype == (torch.uint8 if cfg.pan_rgb else torch.int32)
    for t in (seg_ids, seg_attr):
        assert t.is_cuda and t.dtype == torch.int32 and t.is_contiguous() and tuple(t.shape) == (B, cfg.max_segments)
    assert seg_count.is_cuda and seg_count.dtype == torch.int32 and seg_count.numel() == B
    assert gauss.is_cuda and gauss.dtype == torch.float32 and gauss.numel() == (6 * cfg.sigma + 3) ** 2
    nbytes = ctypes.c_size_t()
    check(lib().mgn_panoptic_targets_workspace_bytes(ctypes.byref(cfg), ctypes.byref(nbytes)), "mgn_panoptic_targets_workspace_bytes")
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
    out = dict(sem_seg=torch.empty((B, H, W), dtype=torch.int64, device=dev),
               center=torch.empty((B, H, W), dtype=torch.float32, device=dev),
               offset=torch.empty((B, 2, H, W), dtype=torch.float32, device=dev),
               sem_seg_weights=torch.empty((B, H, W), dtype=torch.float32, device=dev),
               center_weights=torch.empty((B, 1, H, W), dtype=torch.float32, device=dev),
               offset_weights=torch.empty((B, 1, H, W), dtype=torch.float32, device=dev))
    mask = torch.empty((B, H, W), dtype=torch.bool, device=dev) if want_mask else None
    pts = torch.empty((B, cfg.max_segments, 2), dtype=torch.float64, device=dev) if want_points else None
    area = torch.empty((B, cfg.max_segments), dtype=torch.int64, device=dev) if want_points else None
    opt = lambda t: None if t is None else t.data_ptr()
    check(lib().mgn_panoptic_targets(ctypes.byref(cfg), panoptic.data_ptr(), seg_ids.data_ptr(), seg_attr.data_ptr(),
                                     seg_count.data_ptr(), gauss.data_ptr(), out["sem_seg"].data_ptr(), out["center"].data_ptr(),
                                     out["offset"].data_ptr(), out["sem_seg_weights"].data_ptr(), out["center_weights"].data_ptr(),
                                     out["offset_weights"].data_ptr(), opt(mask), opt(pts), opt(area), ws.data_ptr(), ws.numel(),
                                     _stream()), "mgn_panoptic_targets")
    if want_mask:
        out["reprojection_mask"] = mask
    if want_points:
        out["center_points"], out["seg_area"] = pts, area
    return out


def conv3x3_win(x, w_ohwi, residual=None, patch_rows=16, stats_shift=None, want_stats=False):
    """csrc/conv_win.hip directly (tests / tools; the product reaches it through mgn_conv_igemm's dispatch): 3x3, stride 1, pad 1.
    want_stats: also return the per-patch partial sums [rows, Cout, 2] of (r - shift), (r - shift)^2 over the rounded outputs."""
    N, Cin, H, W = x.shape
    Cout = w_ohwi.shape[0]
    assert w_ohwi.dtype == x.dtype and tuple(w_ohwi.shape[1:]) == (3, 3, Cin)
    out = torch.empty((N, Cout, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    part = None
    if want_stats:
        rows = N * ((H + patch_rows - 1) // patch_rows) * ((W + 31) // 32)
        part = torch.empty((rows, Cout, 2), dtype=torch.float32, device=x.device)
    check(_fn("mgn_conv3x3_win", x)(x.data_ptr(), w_ohwi.data_ptr(), out.data_ptr(), N, H, W, Cin, Cout,
                                    None if residual is None else residual.data_ptr(), patch_rows,
                                    None if part is None else part.data_ptr(), None if stats_shift is None else stats_shift.data_ptr(),
                                    _stream()), "mgn_conv3x3_win")
    return (out, part) if want_stats else out


def conv_up2(dy, w_ihwo, out_hw, residual=None, residual_lowres=False):
    """csrc/conv_up2.hip directly: data gradient of a 3x3 / stride 2 / pad 1 (or 1x1 / stride 2 / pad 0) conv.  dy [N,Cin_k,H,W] 16-bit
    channels_last (gradient of the conv's output), w_ihwo = the layout-mode-1 weights [Cout_k][k][k][Cin_k], out_hw = the conv input's
    (OH, OW).  residual: a full-resolution 16-bit tensor added everywhere, or (residual_lowres) a [N,Cout_k,H,W] tensor added to the even
    pixels.  Returns None when the shape is not one for this kernel (the caller falls back to conv_igemm(up=2))."""
    N, Cin, H, W = dy.shape
    Cout, ks = w_ihwo.shape[0], w_ihwo.shape[1]
    OH, OW = out_hw
    out = torch.empty((N, Cout, OH, OW), dtype=dy.dtype, device=dy.device, memory_format=torch.channels_last)
    rc = _fn("mgn_conv3x3_up2_win", dy)(dy.data_ptr(), w_ihwo.data_ptr(), out.data_ptr(), N, H, W, Cin, Cout, OH, OW, ks,
                                        None if residual is None else residual.data_ptr(), int(bool(residual_lowres)), _stream())
    if rc == -95:   # MGN_ENOTSUP
        return None
    check(rc, "mgn_conv3x3_up2_win")
    return out


def iabn_from_partials(partials, C, M, shift, w32=None, b32=None, eps=1e-5, momentum=0.0, running_mean=None, running_var=None, stats_only=False):
    """Statistics of an activation from the partial sums its producing convolution left behind (mgn_iabn_coeffs_from_partials):
    the coefficient block [4, C] (+ running statistics) like iabn_train_coeffs, or stats [3, C] like iabn_stats."""
    rows = partials.shape[0]
    if rows > 8192:   # the stems: two stages (64 blocks per channel group instead of one)
        stage = torch.empty((64, C, 2), dtype=torch.float32, device=partials.device)
        check(lib().mgn_iabn_partials_reduce(partials.data_ptr(), rows, C, 64, stage.data_ptr(), _stream()), "mgn_iabn_partials_reduce")
        partials, rows = stage, 64
    out = torch.empty((3 if stats_only else 4, C), dtype=torch.float32, device=partials.device)
    p = lambda t: None if t is None else t.data_ptr()
    check(lib().mgn_iabn_coeffs_from_partials(partials.data_ptr(), rows, C, M, p(shift), p(w32), p(b32), eps, momentum,
                                              None if stats_only else p(running_mean), None if stats_only else p(running_var),
                                              None if stats_only else out.data_ptr(), out.data_ptr() if stats_only else None, _stream()),
          "mgn_iabn_coeffs_from_partials")
    return out


def conv_igemm(x, w_ohwi, out_shape, bias, stride, pad, up=1, relu=False, out_dtype=None, khw=None, residual=None, stats=None):
    """x [N,Cin,IH,IW] channels_last bf16; w_ohwi [Cout,KH,KW,Cin] bf16 contiguous (or, for Cin 8/16, the packed
    [Cout, Kpad] layout with khw=(KH,KW)) -> out [N,Cout,OH,OW] channels_last"""
    N, Cin, IH, IW = x.shape
    if khw is None:
        Cout, KH, KW, _ = w_ohwi.shape
    else:
        Cout, (KH, KW) = w_ohwi.shape[0], khw
    OH, OW = out_shape
    out_dtype = x.dtype if out_dtype is None else out_dtype   # the activation format (bf16 / fp16), or fp32
    assert w_ohwi.dtype == x.dtype, "weight layout and activations must share the 16-bit format"
    if stats is not None and up == 1 and bias is None and not relu and residual is None and out_dtype == x.dtype:
        # stats = (shift [Cout] fp32 or None, holder list): a layer whose kernel has the statistics epilogue (windowed 3x3, 64-channel
        # row march) also leaves the partial sums of its output's batch statistics behind (the following InPlaceABNSync then skips its
        # statistics pass); other layers ignore the request
        shifted = ctypes.c_int(0)
        rows = lib().mgn_conv_stat_rows(N, IH, IW, Cin, OH, OW, Cout, KH, KW, stride, pad, ctypes.byref(shifted))
        if rows > 0:
            out = torch.empty((N, Cout, OH, OW), dtype=out_dtype, device=x.device, memory_format=torch.channels_last)
            part = torch.empty((rows, Cout, 2), dtype=torch.float32, device=x.device)
            shift = stats[0] if shifted.value else None
            check(_fn("mgn_conv_igemm_stats", x)(x.data_ptr(), w_ohwi.data_ptr(), out.data_ptr(), N, IH, IW, Cin, OH, OW, Cout, KH, KW, stride, pad,
                                                 part.data_ptr(), None if shift is None else shift.data_ptr(), _stream()), "mgn_conv_igemm_stats")
            stats[1].append((part, shift))
            return out
    out = torch.empty((N, Cout, OH, OW), dtype=out_dtype, device=x.device, memory_format=torch.channels_last)
    check(_fn("mgn_conv_igemm", x)(x.data_ptr(), w_ohwi.data_ptr(), out.data_ptr(), None if bias is None else bias.data_ptr(),
                               N, IH, IW, Cin, OH, OW, Cout, KH, KW, stride, pad, up, int(relu),
                               int(out_dtype == torch.float32), None if residual is None else residual.data_ptr(), _stream()),
          "mgn_conv_igemm")
    return out


# Deferred split-K reductions (engine/reducer.py): while WGRAD_LAZY[0] is set, conv_wgrad(lazy=True) leaves the partial tiles in their
# workspace, returns an UNWRITTEN gradient tensor and registers the reduction under the tensor's address; the gradient reducer
# collects the entries of a bucket's parameters and runs them as ONE launch that writes straight into the bucket (wgrad_reduce_batch)
WGRAD_LAZY = [False]      # False, or the id() of the GradReducer whose backward is running (the owner of the entries it registers)
WGRAD_PENDING = {}        # gradient tensor address -> (descriptor, workspace, shape, owner)


def wgrad_pending_drop(owner):
    """forget the deferred reductions registered by one reducer (its backward raised / a new step starts): entries of ANOTHER reducer in
    the same process -- a second model, a teacher / student pair, bench legs -- are not touched"""
    for k in [k for k, e in WGRAD_PENDING.items() if e[3] == owner]:
        del WGRAD_PENDING[k]
_wgrad_stager = None


def wgrad_reduce_batch(entries):
    """entries: [(desc8 list, workspace tensor, dst tensor)] -> one launch performing every reduction into its dst"""
    global _wgrad_stager
    rows, start = [], 0
    for desc, _ws, dst in entries:
        gy = lib().mgn_conv_wgrad_reduce_blocks(int(desc[2]), int(desc[4]), int(desc[5]))
        assert gy > 0, desc
        rows.append([desc[0], dst.data_ptr(), desc[2], desc[3], desc[4], desc[5], desc[6], desc[7], start, gy])
        start += desc[3] * gy
    table = torch.tensor(rows, dtype=torch.int64)
    dev = entries[0][2].device
    if _wgrad_stager is None:
        _wgrad_stager = PinnedStager(depth=8)
    tdev = _wgrad_stager.stage(table, dev, slot=("wgrad", len(rows)))
    plan_touch(reads=[ws for _d, ws, _v in entries], writes=[dst for _d, _w, dst in entries])
    check(lib().mgn_conv_wgrad_reduce_batch(tdev.data_ptr(), len(rows), start, _stream()), "mgn_conv_wgrad_reduce_batch")
    return tdev


def conv_wgrad(dy, x, kh, kw, stride, pad, cin_real=None, lazy=False):
    """dy [N,Cout,OH,OW], x [N,Cin,IH,IW] (channels_last bf16) -> dw fp32 in the torch parameter layout
    [Cout, cin_real, KH, KW] (cin_real < Cin for the channel-padded stem inputs)"""
    N, Cout, OH, OW = dy.shape
    _, Cin, IH, IW = x.shape
    cin_real = Cin if cin_real is None else cin_real
    dw = torch.empty((Cout, cin_real, kh, kw), dtype=torch.float32, device=x.device)
    nb = ctypes.c_size_t(0)
    check(lib().mgn_conv_wgrad_workspace_bytes(N, OH, OW, Cin, Cout, kh, kw, ctypes.byref(nb)), "mgn_conv_wgrad_workspace_bytes")
    ws = torch.empty(nb.value, dtype=torch.uint8, device=x.device)
    if lazy and WGRAD_LAZY[0]:
        desc = (ctypes.c_longlong * 8)()
        rc = _fn("mgn_conv_wgrad_partial", dy)(dy.data_ptr(), x.data_ptr(), N, IH, IW, Cin, OH, OW, Cout, kh, kw, stride, pad, cin_real,
                                               ws.data_ptr(), nb.value, desc, _stream())
        if rc == 0:
            WGRAD_PENDING[dw.data_ptr()] = (list(desc), ws, tuple(dw.shape), WGRAD_LAZY[0])
            return dw
        if rc != -95:
            check(rc, "mgn_conv_wgrad_partial")
    check(_fn("mgn_conv_wgrad", dy)(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), N, IH, IW, Cin, OH, OW, Cout, kh, kw, stride, pad,
                               cin_real, ws.data_ptr(), nb.value, _stream()), "mgn_conv_wgrad")
    return dw


def conv_igemm_act(x, w_ohwi, out_shape, bias, stride, pad, act=0, slope=0.0, residual=None):
    """act(conv(x, w) + bias + residual) in one launch (mgn_conv_igemm_act; act 0 none | 1 ReLU | 2 leaky ReLU): the inference form of
    conv -> InPlaceABNSync [-> + shortcut -> ReLU] with the norm folded into w_ohwi / bias.  None where no kernel has the epilogue."""
    N, Cin, IH, IW = x.shape
    Cout, KH, KW, _ = w_ohwi.shape
    OH, OW = out_shape
    out = torch.empty((N, Cout, OH, OW), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    rc = _fn("mgn_conv_igemm_act", x)(x.data_ptr(), w_ohwi.data_ptr(), out.data_ptr(), None if bias is None else bias.data_ptr(), N, IH, IW, Cin,
                                      OH, OW, Cout, KH, KW, stride, pad, act, float(slope), None if residual is None else residual.data_ptr(), _stream())
    if rc == -95:
        return None
    check(rc, "mgn_conv_igemm_act")
    return out


def conv1x1_cat(a, b, w_ohwi):
    """conv1x1(cat([a, b], 1), w) without the concatenated map: a, b [N,128,H,W] channels_last 16-bit, w_ohwi [Cout,1,1,256]; None if the
    shape is not one the two-source streaming kernel takes (mgn_conv1x1_cat)"""
    N, Ca, H, W = a.shape
    Cout = w_ohwi.shape[0]
    if not (Ca == 128 and b.shape == a.shape and Cout % 256 == 0 and a.dtype == b.dtype == w_ohwi.dtype and w_ohwi.shape[1:] == (1, 1, 256)):
        return None
    out = torch.empty((N, Cout, H, W), dtype=a.dtype, device=a.device, memory_format=torch.channels_last)
    rc = _fn("mgn_conv1x1_cat", a)(a.data_ptr(), b.data_ptr(), w_ohwi.data_ptr(), out.data_ptr(), N, H, W, 256, Cout, _stream())
    if rc == -95:
        return None
    check(rc, "mgn_conv1x1_cat")
    return out


def conv1x1_split(dy, w_ihwo):
    """the data gradient of conv1x1_cat: dy [N,256,H,W], w_ihwo [256,1,1,256] (layout mode 1) -> (da, db) [N,128,H,W] each, or None"""
    N, C, H, W = dy.shape
    if not (C == 256 and tuple(w_ihwo.shape) == (256, 1, 1, 256) and dy.dtype == w_ihwo.dtype):
        return None
    da = torch.empty((N, 128, H, W), dtype=dy.dtype, device=dy.device, memory_format=torch.channels_last)
    db = torch.empty_like(da)
    rc = _fn("mgn_conv1x1_split", dy)(dy.data_ptr(), w_ihwo.data_ptr(), da.data_ptr(), db.data_ptr(), N, H, W, 256, 256, _stream())
    if rc == -95:
        return None
    check(rc, "mgn_conv1x1_split")
    return da, db


def conv_wgrad_cat(dy, a, b, lazy=False):
    """weight gradient of conv1x1_cat: dy [N,Cout,H,W], a / b [N,C/2,H,W] -> dw fp32 [Cout, C, 1, 1] (lazy: as conv_wgrad)"""
    N, Cout, H, W = dy.shape
    Cin = a.shape[1] + b.shape[1]
    dw = torch.empty((Cout, Cin, 1, 1), dtype=torch.float32, device=dy.device)
    nb = ctypes.c_size_t(0)
    check(lib().mgn_conv_wgrad_workspace_bytes(N, H, W, Cin, Cout, 1, 1, ctypes.byref(nb)), "mgn_conv_wgrad_workspace_bytes")
    ws = torch.empty(nb.value, dtype=torch.uint8, device=dy.device)
    if lazy and WGRAD_LAZY[0]:
        desc = (ctypes.c_longlong * 8)()
        check(_fn("mgn_conv_wgrad_cat", dy)(dy.data_ptr(), a.data_ptr(), b.data_ptr(), None, N, H, W, Cin, Cout, ws.data_ptr(), nb.value, desc, _stream()),
              "mgn_conv_wgrad_cat")
        WGRAD_PENDING[dw.data_ptr()] = (list(desc), ws, tuple(dw.shape), WGRAD_LAZY[0])
        return dw
    check(_fn("mgn_conv_wgrad_cat", dy)(dy.data_ptr(), a.data_ptr(), b.data_ptr(), dw.data_ptr(), N, H, W, Cin, Cout, ws.data_ptr(), nb.value, None,
                                         _stream()), "mgn_conv_wgrad_cat")
    return dw


def _layout_empty(w, mode, Cp, cout_pad=0, dtype=torch.bfloat16):
    Cout, Cin, KH, KW = w.shape
    Cout = max(Cout, cout_pad)
    if mode == 0:
        return torch.empty((Cout, KH, KW, Cin), dtype=dtype, device=w.device)
    if mode == 1:
        return torch.empty((Cin, KH, KW, Cout), dtype=dtype, device=w.device)
    return torch.empty((Cout, (KH * KW * Cp + 31) // 32 * 32), dtype=dtype, device=w.device)


def weight_layout(w, mode, Cp=0, cout_pad=0, dtype=torch.bfloat16):
    """fp32 OIHW parameter -> bf16 kernel layout (0: OHWI, 1: flipped IHWO for the data gradient, 2: packed stem), output
    channels zero-padded to `cout_pad`.  Parameters are served from `weight_cache` (all layouts refreshed by ONE launch
    after the optimizer step)."""
    return weight_cache.get(w, mode, Cp, cout_pad, dtype)


def _weight_layout_now(w, mode, Cp, out=None, cout_pad=0, dtype=torch.bfloat16):
    Cout, Cin, KH, KW = w.shape
    out = _layout_empty(w, mode, Cp, cout_pad, dtype) if out is None else out
    wc = w.detach()
    wc = wc if (wc.dtype == torch.float32 and wc.is_contiguous()) else wc.float().contiguous()
    check(_fn("mgn_weight_layout", out)(wc.data_ptr(), out.data_ptr(), Cout, Cin, KH, KW, mode, Cp, cout_pad, _stream()), "mgn_weight_layout")
    return out


class _WeightCache:
    """bf16 kernel layouts of the conv parameters.  A layout is valid while the parameter's storage and version are
    unchanged and no `refresh()` happened since; `refresh()` (called by FusedAdam.step, whose kernel updates the flat
    parameter buffer behind torch's version counter) re-derives EVERY registered layout with one batched launch."""

    def __init__(self):
        self.entries = {}     # (id(param), mode, Cp) -> dict(ref, out, version, ptr)
        self.tables = {}      # dtype -> (device table of the batched kernel, blocks); rebuilt when the entry set changed
        self.table = []
        self.dirty = True
        self.off = bool(os.environ.get("MGN_NO_WCACHE"))   # A/B switch: convert per call

    def get(self, w, mode, Cp=0, cout_pad=0, dtype=torch.bfloat16):
        import weakref
        if self.off or not (isinstance(w, torch.nn.Parameter) and w.is_leaf and w.dtype == torch.float32 and w.is_contiguous()):
            return _weight_layout_now(w, mode, Cp, None, cout_pad, dtype)   # temporaries: converted per call
        key = (id(w), mode, Cp, cout_pad, dtype)
        e = self.entries.get(key)
        if e is not None and e["ref"]() is w and e["ptr"] == w.data_ptr() and e["version"] == w._version:
            return e["out"]
        out = _weight_layout_now(w, mode, Cp, None if e is None or e["ref"]() is not w else e["out"], cout_pad, dtype)
        if e is None or e["ref"]() is not w or e["ptr"] != w.data_ptr():
            self.dirty = True
        self.entries[key] = dict(ref=weakref.ref(w), out=out, version=w._version, ptr=w.data_ptr(), mode=mode, Cp=Cp, cout_pad=cout_pad, dtype=dtype)
        return out

    def _rebuild(self):
        """one table of the batched kernel per 16-bit format (bf16 entries are converted by mgn_weight_layout_batch, fp16 entries
        by its _f16 twin)"""
        dead = [k for k, e in self.entries.items() if e["ref"]() is None]
        for k in dead:
            del self.entries[k]
        self.tables = {}
        dev = None
        for dtype in H16:
            rows, blocks = [], 0
            for e in self.entries.values():
                w = e["ref"]()
                if w is None or e["dtype"] != dtype:   # (collected since the filter above: a cyclic-garbage pass can run at any allocation)
                    continue
                if dev is None:
                    dev = w.device
                if w.device != dev:
                    continue
                Cout, Cin, KH, KW = w.shape
                # work items of the batched kernel: (co, ci) pairs for the plain layouts, output elements for the packed stems
                n_items = e["out"].numel() if e["mode"] == 2 else max(Cout, e["cout_pad"]) * Cin
                rows.append([w.data_ptr(), e["out"].data_ptr(), n_items, blocks, Cout | (e["cout_pad"] << 32), Cin, (KH << 32) | KW,
                             (e["mode"] << 32) | e["Cp"]])
                e["ptr"] = w.data_ptr()
                blocks += (n_items + 255) // 256
            if rows:
                self.tables[dtype] = (torch.tensor(rows, dtype=torch.int64).to(dev), blocks)
        self.table = [t for t, _ in self.tables.values()]   # (kept alive by a captured graph, see Trainer.capture_step)
        self.dirty = False

    def refresh(self):
        """re-derive every registered layout from the current parameter values (one launch per 16-bit format in use)"""
        self.epoch = getattr(self, "epoch", 0) + 1   # (parameters changed behind torch's version counter: ops.conv_abn_eval's folds expire)
        for e in self.entries.values():
            w = e["ref"]()
            if w is None or e["ptr"] != w.data_ptr():   # a collected parameter's row must leave the table (its memory is gone)
                self.dirty = True
                break
        if self.dirty:
            self._rebuild()
        for dtype, (table, blocks) in self.tables.items():
            if PLAN_RECORDER[0] is not None:
                ents = [e for e in self.entries.values() if e["dtype"] == dtype and e["ref"]() is not None]
                plan_touch(reads=[e["ref"]() for e in ents], writes=[e["out"] for e in ents])
            check(_fn("mgn_weight_layout_batch", dtype)(table.data_ptr(), table.shape[0], blocks, _stream()), "mgn_weight_layout_batch")
        for e in self.entries.values():
            w = e["ref"]()
            if w is not None:
                e["version"] = w._version


weight_cache = _WeightCache()


# ---------------------------------------------------------------------------------------------------------------
# head losses fused with the bilinear upsampling
# ---------------------------------------------------------------------------------------------------------------
def _lr_strides(t):
    """(sb, sh, sw) element strides of a logical [B,C,h,w] low-res map whose channel stride is 1"""
    assert t.stride(1) == 1 or t.shape[1] == 1, "low-res map must be channels-last"
    return t.stride(0), t.stride(2), t.stride(3)


def upce_supported(lr):
    sb, sh, sw = lr.stride(0), lr.stride(2), lr.stride(3)
    return (lr.is_cuda and lr.dtype in H16 and lr.stride(1) == 1 and lr.shape[1] <= 32 and sb % 8 == 0 and sh % 8 == 0
            and sw % 8 == 0 and sw >= (lr.shape[1] + 7) // 8 * 8 and lr.shape[2] >= 2 and lr.shape[3] >= 2)


def upce_fwd(lr, labels, weights, H, W, ignore, thr):
    B, K, h, w = lr.shape
    sb, sh, sw = _lr_strides(lr)
    ce = torch.empty((B, H, W), dtype=torch.float32, device=lr.device)
    partials = torch.empty(lib().mgn_upce_partials(B, H, W) * 3, dtype=torch.float32, device=lr.device)
    sums = torch.empty(3, dtype=torch.float32, device=lr.device)
    check(_fn("mgn_upce_fwd", lr)(lr.data_ptr(), sb, sh, sw, B, h, w, H, W, K, labels.data_ptr(),
                             None if weights is None else weights.data_ptr(), ignore, thr, ce.data_ptr(), partials.data_ptr(),
                             sums.data_ptr(), _stream()), "mgn_upce_fwd")
    return ce, sums


def ohem_select(ce, sums, thr, n_sel, force_topk):
    """device-side OHEM / top-k selection -> (sel3, loss) without a host synchronisation"""
    n = ce.numel()
    nb = ctypes.c_size_t(0)
    check(lib().mgn_ohem_select_workspace_bytes(n, ctypes.byref(nb)), "mgn_ohem_select_workspace_bytes")
    ws = torch.empty(nb.value, dtype=torch.uint8, device=ce.device)
    out = torch.empty(4, dtype=torch.float32, device=ce.device)
    check(lib().mgn_ohem_select(ce.data_ptr(), n, sums.data_ptr(), float(thr), int(n_sel), int(force_topk), out.data_ptr(),
                                out[3:].data_ptr(), ws.data_ptr(), nb.value, _stream()), "mgn_ohem_select")
    return out[:3], out[3]


def _adjoint_dst(which, shape, dev, B, h, w, H, W, channels):
    """destination + footprint table of the tile-wise bilinear adjoints (csrc/headloss.hip).  Default: the reproducible two-kernel
    form (every tile stores its footprint, a gather sums them in a fixed order: uninitialised destination); MGN_ADJOINT_ATOMICS=1:
    float atomics into a zeroed destination (order-dependent last bits; MGN_SERIAL_SCATTER=1 orders them, one tile per launch)."""
    if os.environ.get("MGN_ADJOINT_ATOMICS") or os.environ.get("MGN_SERIAL_SCATTER"):
        return torch.zeros(shape, dtype=torch.float32, device=dev), None
    n = ctypes.c_size_t(0)
    check(lib().mgn_adjoint_footprint_floats(which, B, h, w, H, W, channels, ctypes.byref(n)), "mgn_adjoint_footprint_floats")
    return torch.empty(shape, dtype=torch.float32, device=dev), torch.empty(n.value, dtype=torch.float32, device=dev)


def upce_bwd(lr, labels, weights, H, W, ignore, ce, sel3, gout, Kp):
    B, K, h, w = lr.shape
    sb, sh, sw = _lr_strides(lr)
    dlg, foot = _adjoint_dst(0, (B, h, w, Kp), lr.device, B, h, w, H, W, K)
    check(_fn("mgn_upce_bwd", lr)(lr.data_ptr(), sb, sh, sw, B, h, w, H, W, K, Kp, labels.data_ptr(),
                             None if weights is None else weights.data_ptr(), ignore, ce.data_ptr(), sel3.data_ptr(),
                             gout.data_ptr(), dlg.data_ptr(), None if foot is None else foot.data_ptr(), _stream()), "mgn_upce_bwd")
    return dlg


def ins_loss_fwd(center_lr, offset_lr, H, W, ct, cw, ot, ow, oscale):
    B, _, h, w = center_lr.shape
    partials = torch.empty(lib().mgn_upce_partials(B, H, W) * 4, dtype=torch.float32, device=ct.device)
    out4 = torch.empty(4, dtype=torch.float32, device=ct.device)
    cs, os_ = _lr_strides(center_lr), _lr_strides(offset_lr)
    check(_fn("mgn_ins_loss_fwd", offset_lr)(center_lr.data_ptr(), *cs, offset_lr.data_ptr(), *os_, B, h, w, H, W, ct.data_ptr(), cw.data_ptr(),
                                 ot.data_ptr(), ow.data_ptr(), oscale, partials.data_ptr(), out4.data_ptr(), _stream()),
          "mgn_ins_loss_fwd")
    return out4


def ins_loss_bwd(center_lr, offset_lr, H, W, ct, cw, ot, ow, oscale, out4, gout2):
    B, _, h, w = center_lr.shape
    dco, foot = _adjoint_dst(1, (B, h, w, 4), ct.device, B, h, w, H, W, 3)
    cs, os_ = _lr_strides(center_lr), _lr_strides(offset_lr)
    check(_fn("mgn_ins_loss_bwd", offset_lr)(center_lr.data_ptr(), *cs, offset_lr.data_ptr(), *os_, B, h, w, H, W, ct.data_ptr(), cw.data_ptr(),
                                 ot.data_ptr(), ow.data_ptr(), oscale, out4.data_ptr(), gout2.data_ptr(), dco.data_ptr(),
                                 None if foot is None else foot.data_ptr(), _stream()),
          "mgn_ins_loss_bwd")
    return dco


def upsample1_fwd(lr, H, W):
    B, _, h, w = lr.shape
    out = torch.empty((B, 1, H, W), dtype=torch.float32, device=lr.device)
    check(lib().mgn_upsample1_fwd(lr.data_ptr(), B, h, w, H, W, out.data_ptr(), _stream()), "mgn_upsample1_fwd")
    return out


def upsample1_bwd(dfull, h, w):
    B, _, H, W = dfull.shape
    dlr, foot = _adjoint_dst(2, (B, 1, h, w), dfull.device, B, h, w, H, W, 1)
    check(lib().mgn_upsample1_bwd(dfull.data_ptr(), B, h, w, H, W, dlr.data_ptr(), None if foot is None else foot.data_ptr(), _stream()),
          "mgn_upsample1_bwd")
    return dlr


def maxpool_fwd(x):
    N, C, IH, IW = x.shape
    OH, OW = (IH - 1) // 2 + 1, (IW - 1) // 2 + 1
    y = torch.empty((N, C, OH, OW), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    arg = torch.empty((N, OH, OW, C), dtype=torch.uint8, device=x.device)
    check(_fn("mgn_maxpool3x3s2_fwd", x)(x.data_ptr(), y.data_ptr(), arg.data_ptr(), N, IH, IW, C, _stream()), "mgn_maxpool3x3s2_fwd")
    return y, arg


def abn_maxpool_fwd(x, scale, offset, activation, slope):
    """x [N,C,IH,IW] bf16 channels_last (conv output, left untouched) -> max_pool3x3s2(act(scale*x+offset)), argmax"""
    N, C, IH, IW = x.shape
    OH, OW = (IH - 1) // 2 + 1, (IW - 1) // 2 + 1
    y = torch.empty((N, C, OH, OW), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    arg = torch.empty((N, OH, OW, C), dtype=torch.uint8, device=x.device)
    check(_fn("mgn_abn_maxpool_fwd", x)(x.data_ptr(), scale.data_ptr(), offset.data_ptr(), activation, slope, y.data_ptr(),
                                    arg.data_ptr(), N, IH, IW, C, _stream()), "mgn_abn_maxpool_fwd")
    return y, arg


def abn_maxpool_bwd(x, dpool, arg, coef, weight, bias, sums, total_count, eps, activation, slope):
    N, C, IH, IW = x.shape
    dx = torch.empty_like(x)
    check(_fn("mgn_abn_maxpool_bwd", x)(x.data_ptr(), dpool.data_ptr(), arg.data_ptr(), dx.data_ptr(), coef[0].data_ptr(),
                                    coef[1].data_ptr(), weight.data_ptr(), bias.data_ptr(), coef[3].data_ptr(), sums.data_ptr(),
                                    float(total_count), eps, activation, slope, N, IH, IW, C, _stream()), "mgn_abn_maxpool_bwd")
    return dx


def maxpool_bwd(dy, arg, in_shape):
    N, C, IH, IW = in_shape
    dx = torch.empty((N, C, IH, IW), dtype=dy.dtype, device=dy.device, memory_format=torch.channels_last)
    check(_fn("mgn_maxpool3x3s2_bwd", dy)(dy.data_ptr(), arg.data_ptr(), dx.data_ptr(), N, IH, IW, C, _stream()), "mgn_maxpool3x3s2_bwd")
    return dx


# ---------------------------------------------------------------------------------------------------------------
# element-wise / broadcast / pooling glue (channels-last bf16)
# ---------------------------------------------------------------------------------------------------------------
def elt_supported(x):
    C = x.shape[1]
    return (x.is_cuda and x.dtype in H16 and x.dim() == 4 and C % 8 == 0 and C // 8 <= 256 and 256 % (C // 8) == 0
            and x.is_contiguous(memory_format=torch.channels_last))


def _cl_like(x, shape=None):
    return torch.empty(tuple(x.shape) if shape is None else shape, dtype=x.dtype, device=x.device, memory_format=torch.channels_last)


def sum3(a, b, c=None):
    """a + b (+ c) of 16-bit tensors of the same layout (fp32 sum, one rounding): csrc/eltwise.hip"""
    y = torch.empty_like(a)
    check(_fn("mgn_sum3", a)(a.data_ptr(), b.data_ptr(), None if c is None else c.data_ptr(), y.data_ptr(), a.numel(), _stream()), "mgn_sum3")
    return y


def add_relu_fwd(a, b):
    y = _cl_like(a)
    check(_fn("mgn_add_relu_fwd", a)(a.data_ptr(), b.data_ptr(), y.data_ptr(), a.numel(), _stream()), "mgn_add_relu_fwd")
    return y


def relu_mask_bwd(dy, y):
    dx = _cl_like(y)
    check(_fn("mgn_relu_mask_bwd", dy)(dy.data_ptr(), y.data_ptr(), dx.data_ptr(), y.numel(), _stream()), "mgn_relu_mask_bwd")
    return dx


def colsum(x, x2, scale):
    N, C, H, W = x.shape
    out = torch.empty((N, C), dtype=torch.float32, device=x.device)
    ws = torch.empty(N * 64 * C, dtype=torch.float32, device=x.device)
    check(_fn("mgn_colsum", x)(x.data_ptr(), None if x2 is None else x2.data_ptr(), N, H * W, C, scale, out.data_ptr(), ws.data_ptr(),
                           ws.numel() * 4, _stream()), "mgn_colsum")
    return out


def abn_apply_pool(x, z, scale, offset, activation, slope, pool_scale):
    """z = act(scale * x + offset) (z may be x) and pooled [N, C] = pool_scale * column sums of the rounded z per image, in ONE pass"""
    N, C, H, W = x.shape
    pooled = torch.empty((N, C), dtype=torch.float32, device=x.device)
    ws = torch.empty(N * 256 * C, dtype=torch.float32, device=x.device)
    check(_fn("mgn_abn_apply_pool", x)(x.data_ptr(), z.data_ptr(), scale.data_ptr(), offset.data_ptr(), activation, slope, N, H * W, C, pool_scale,
                                       pooled.data_ptr(), ws.data_ptr(), ws.numel() * 4, _stream()), "mgn_abn_apply_pool")
    return pooled


def att_abn_bwd_stats(g, z, weight, bias, eps, activation, slope):
    """S [5, N, C]: per-(image, channel) sums {g z, g m, g m xh, m, m xh} of the fused attention + norm backward (csrc/eltwise.hip)"""
    N, C, H, W = z.shape
    S = torch.empty((5, N, C), dtype=torch.float32, device=z.device)
    ws = torch.empty(N * 64 * 5 * C, dtype=torch.float32, device=z.device)
    check(_fn("mgn_att_abn_bwd_stats", z)(g.data_ptr(), z.data_ptr(), weight.data_ptr(), bias.data_ptr(), eps, activation, slope, N, H * W, C,
                                          S.data_ptr(), ws.data_ptr(), ws.numel() * 4, _stream()), "mgn_att_abn_bwd_stats")
    return S


def att_abn_bwd_sums(S, s, dpool, mode, weight):
    """-> (sums [2, C], d_weight [C], d_bias [C]) of the norm's backward from the per-image sums, the attention factor and the pooled gradient"""
    _, N, C = S.shape
    out = torch.empty((4, C), dtype=torch.float32, device=S.device)
    check(lib().mgn_att_abn_bwd_sums(S.data_ptr(), s.data_ptr(), None if dpool is None else dpool.data_ptr(), mode, N, C, weight.data_ptr(),
                                     out.data_ptr(), out[2].data_ptr(), _stream()), "mgn_att_abn_bwd_sums")
    return out[:2], out[2], out[3]


def att_abn_bwd_apply(g, z, s, dpool, mode, weight, bias, rstd, sums, total_count, eps, activation, slope):
    N, C, H, W = z.shape
    dy = _cl_like(z)
    check(_fn("mgn_att_abn_bwd_apply", z)(g.data_ptr(), z.data_ptr(), dy.data_ptr(), s.data_ptr(), None if dpool is None else dpool.data_ptr(), mode,
                                          weight.data_ptr(), bias.data_ptr(), rstd.data_ptr(), sums.data_ptr(), 1.0 / float(total_count), eps,
                                          activation, slope, N, H * W, C, _stream()), "mgn_att_abn_bwd_apply")
    return dy


def colsum_all(x):
    """[C] fp32 = sum over N, H, W of a 16-bit channels_last tensor (mgn_colsum with the batch folded into the rows)"""
    N, C, H, W = x.shape
    out = torch.empty((1, C), dtype=torch.float32, device=x.device)
    ws = torch.empty(64 * C, dtype=torch.float32, device=x.device)
    check(_fn("mgn_colsum", x)(x.data_ptr(), None, 1, N * H * W, C, 1.0, out.data_ptr(), ws.data_ptr(), ws.numel() * 4, _stream()), "mgn_colsum")
    return out[0]


def bcast_rows(g, shape, scale, dtype=torch.bfloat16):
    N, C, H, W = shape
    dx = torch.empty(shape, dtype=dtype, device=g.device, memory_format=torch.channels_last)
    check(_fn("mgn_bcast_rows", dx)(g.data_ptr(), N, H * W, C, scale, dx.data_ptr(), _stream()), "mgn_bcast_rows")
    return dx


def scale_channels(x, s, mode, add=None, addt=None):
    """x * s[n,c] (mode 0) | x * (1 + s[n,c]) (mode 1)  [+ add[n,c]]  [+ addt: a 16-bit tensor of x's shape and layout]"""
    N, C, H, W = x.shape
    y = _cl_like(x)
    assert addt is None or (addt.shape == x.shape and addt.dtype == x.dtype and addt.is_contiguous(memory_format=torch.channels_last))
    check(_fn("mgn_scale_channels", x)(x.data_ptr(), s.data_ptr(), N, H * W, C, mode, None if add is None else add.data_ptr(),
                                   None if addt is None else addt.data_ptr(), y.data_ptr(), _stream()), "mgn_scale_channels")
    return y


_ACT = {None: 0, "none": 0, "relu": 1, "sigmoid": 2}


def vec_linear_fwd(v, w, act, bn=None):
    """v [N,K] fp32, w [C,K(,1,1)] fp32 -> act(bn(v @ w.T)).  bn = (weight, bias, running_mean, running_var, training, momentum, eps)
    -> (out [N,C], xhat | None, rstd | None)"""
    N, K = v.shape
    C = w.shape[0]
    out = torch.empty((N, C), dtype=torch.float32, device=v.device)
    xhat = rstd = None
    if bn is not None:
        bw, bb, rm, rv, training, momentum, eps = bn
        if training:
            xhat, rstd = torch.empty_like(out), torch.empty(C, dtype=torch.float32, device=v.device)
        check(lib().mgn_vec_linear_fwd(v.data_ptr(), w.data_ptr(), N, K, C, _ACT[act], bw.data_ptr(), bb.data_ptr(),
                                       None if rm is None else rm.data_ptr(), None if rv is None else rv.data_ptr(), int(training),
                                       momentum, eps, out.data_ptr(), None if xhat is None else xhat.data_ptr(),
                                       None if rstd is None else rstd.data_ptr(), _stream()), "mgn_vec_linear_fwd")
    else:
        check(lib().mgn_vec_linear_fwd(v.data_ptr(), w.data_ptr(), N, K, C, _ACT[act], None, None, None, None, 0, 0.0, 0.0,
                                       out.data_ptr(), None, None, _stream()), "mgn_vec_linear_fwd")
    return out, xhat, rstd


def vec_linear_bwd(dout, out, v, w, act, bn_weight=None, xhat=None, rstd=None, eps=0.0, dv_scale=1.0):
    """-> (dW like w, dv [N,K] * dv_scale, d bn weight, d bn bias)"""
    N, K = v.shape
    C = w.shape[0]
    dW = torch.empty_like(w)
    dv = torch.empty_like(v)
    dbw = dbb = None
    if bn_weight is not None:
        dbw, dbb = torch.empty_like(bn_weight), torch.empty_like(bn_weight)
    nb = ctypes.c_size_t(0)
    check(lib().mgn_vec_linear_bwd_workspace_bytes(N, K, C, ctypes.byref(nb)), "mgn_vec_linear_bwd_workspace_bytes")
    ws = torch.empty(nb.value, dtype=torch.uint8, device=v.device)
    check(lib().mgn_vec_linear_bwd(dout.data_ptr(), out.data_ptr(), v.data_ptr(), w.data_ptr(), N, K, C, _ACT[act],
                                   None if bn_weight is None else bn_weight.data_ptr(), None if xhat is None else xhat.data_ptr(),
                                   None if rstd is None else rstd.data_ptr(), eps, dv_scale, dW.data_ptr(), dv.data_ptr(),
                                   None if dbw is None else dbw.data_ptr(), None if dbb is None else dbb.data_ptr(), ws.data_ptr(),
                                   nb.value, _stream()), "mgn_vec_linear_bwd")
    return dW, dv, dbw, dbb


def nearest_fwd(x, H, W):
    N, C, h, w = x.shape
    y = _cl_like(x, (N, C, H, W))
    check(_fn("mgn_nearest_fwd", x)(x.data_ptr(), N, h, w, H, W, C, y.data_ptr(), _stream()), "mgn_nearest_fwd")
    return y


def nearest_bwd(dy, h, w):
    N, C, H, W = dy.shape
    dx = _cl_like(dy, (N, C, h, w))
    check(_fn("mgn_nearest_bwd", dy)(dy.data_ptr(), N, h, w, H, W, C, dx.data_ptr(), _stream()), "mgn_nearest_bwd")
    return dx


def concat2(a, b):
    N, Ca, H, W = a.shape
    Cb = b.shape[1]
    y = _cl_like(a, (N, Ca + Cb, H, W))
    check(lib().mgn_concat2(a.data_ptr(), b.data_ptr(), N * H * W, Ca, Cb, y.data_ptr(), _stream()), "mgn_concat2")
    return y


def split2(dy, Ca, Cb):
    N, _, H, W = dy.shape
    da, db = _cl_like(dy, (N, Ca, H, W)), _cl_like(dy, (N, Cb, H, W))
    check(lib().mgn_split2(dy.data_ptr(), N * H * W, Ca, Cb, da.data_ptr(), db.data_ptr(), _stream()), "mgn_split2")
    return da, db


# ---------------------------------------------------------------------------------------------------------------
# stand-alone geometry stages (csrc/geometry.hip)
# ---------------------------------------------------------------------------------------------------------------
def _geo_partials(B, H, W, device):
    n = ctypes.c_size_t(0)
    check(lib().mgn_geometry_partial_rows(B, H, W, ctypes.byref(n)), "mgn_geometry_partial_rows")
    return torch.empty((B, n.value // B, 12), dtype=torch.float32, device=device)


def _geo_sum(partials):
    """[B, rows, 12] block partials -> (dA [B,3,3], dt [B,3]); fp64 accumulation, fixed order"""
    s = partials.double().sum(1).float()
    return s[:, :9].reshape(-1, 3, 3), s[:, 9:]


def view_synthesis_fwd(ref, depth, A, t, padding_mode="zeros"):
    B, C, H, W = ref.shape
    out = torch.empty_like(ref)
    check(lib().mgn_view_synthesis_fwd(_dev_f32(ref, "ref_image").data_ptr(), _dev_f32(depth, "depth").data_ptr(),
                                       _dev_f32(A, "A").data_ptr(), _dev_f32(t, "t").data_ptr(), B, C, H, W, _PAD[padding_mode],
                                       out.data_ptr(), _stream()), "mgn_view_synthesis_fwd")
    return out


def view_synthesis_bwd(ref, depth, A, t, g_out, padding_mode="zeros"):
    B, C, H, W = ref.shape
    d_depth = torch.empty_like(depth)
    part = _geo_partials(B, H, W, ref.device)
    check(lib().mgn_view_synthesis_bwd(ref.data_ptr(), depth.data_ptr(), A.data_ptr(), t.data_ptr(),
                                       _dev_f32(g_out, "grad").data_ptr(), B, C, H, W, _PAD[padding_mode], d_depth.data_ptr(),
                                       part.data_ptr(), _stream()), "mgn_view_synthesis_bwd")
    return (d_depth,) + _geo_sum(part)


def reconstruct_fwd(depth, A, t):
    B, _, H, W = depth.shape
    pts = torch.empty((B, 3, H, W), dtype=torch.float32, device=depth.device)
    check(lib().mgn_reconstruct_fwd(_dev_f32(depth, "depth").data_ptr(), _dev_f32(A, "A").data_ptr(), _dev_f32(t, "t").data_ptr(),
                                    B, H, W, pts.data_ptr(), _stream()), "mgn_reconstruct_fwd")
    return pts


def reconstruct_bwd(depth, A, t, g):
    B, _, H, W = depth.shape
    d_depth = torch.empty_like(depth)
    part = _geo_partials(B, H, W, depth.device)
    check(lib().mgn_reconstruct_bwd(depth.data_ptr(), A.data_ptr(), t.data_ptr(), _dev_f32(g, "grad").data_ptr(), B, H, W,
                                    d_depth.data_ptr(), part.data_ptr(), _stream()), "mgn_reconstruct_bwd")
    return (d_depth,) + _geo_sum(part)


def project_fwd(points, A, t):
    B, _, H, W = points.shape
    coords = torch.empty((B, H, W, 2), dtype=torch.float32, device=points.device)
    check(lib().mgn_project_fwd(_dev_f32(points, "points").data_ptr(), _dev_f32(A, "A").data_ptr(), _dev_f32(t, "t").data_ptr(),
                                B, H, W, coords.data_ptr(), _stream()), "mgn_project_fwd")
    return coords


def project_bwd(points, A, t, g):
    B, _, H, W = points.shape
    d_pts = torch.empty_like(points)
    part = _geo_partials(B, H, W, points.device)
    check(lib().mgn_project_bwd(points.data_ptr(), A.data_ptr(), t.data_ptr(), _dev_f32(g, "grad").data_ptr(), B, H, W,
                                d_pts.data_ptr(), part.data_ptr(), _stream()), "mgn_project_bwd")
    return (d_pts,) + _geo_sum(part)
