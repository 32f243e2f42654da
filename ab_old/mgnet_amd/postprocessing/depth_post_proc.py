"""Depth post-processing -- host-side mirror of mgnet/postprocessing/depth_post_proc.py:11-70 (same function name,
arguments, assertions and results) over `mgn_depth_post` (csrc/postproc.hip)."""
import torch

from .. import _C

__all__ = ["get_depth_prediction"]


def get_depth_prediction(depth_logits, use_dgc_scaling, camera_matrix=None, real_camera_height=None, panoptic_seg=None,
                         road_class_id=-1, depth_filter_class_ids=None):
    """depth_logits [1,1,H,W] -> (depth [H,W], cam_xyz_points [3,H,W] or None).  The input is left untouched (the
    reference rescales it in place)."""
    if use_dgc_scaling:
        assert camera_matrix is not None, "camera_matrix is necessary for dgc rescaling!"
        assert real_camera_height is not None, "real_camera_height is necessary for dgc rescaling!"
        if panoptic_seg is not None:
            assert road_class_id != -1, "road_class_id is necessary for dgc rescaling using panoptic prediction!"
    if not depth_logits.is_cuda:
        raise RuntimeError("get_depth_prediction runs on the GPU (no CPU fallback by design)")
    H, W = depth_logits.shape[-2:]
    ids = list(depth_filter_class_ids or []) if panoptic_seg is not None else []
    if len(ids) > _C.DEPTH_MAX_FILTER_IDS:
        raise ValueError(f"at most {_C.DEPTH_MAX_FILTER_IDS} depth_filter_class_ids")
    fx = fy = 1.0
    cx = cy = 0.0
    h = 1.0
    if use_dgc_scaling:
        K = torch.as_tensor(camera_matrix, dtype=torch.float32).reshape(-1, camera_matrix.shape[-1])[:3].cpu()
        fx, fy, cx, cy = float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])
        h = float(torch.as_tensor(real_camera_height, dtype=torch.float32).reshape(-1)[0])
    cfg = _C.DepthPostCfg(H, W, int(bool(use_dgc_scaling)), int(panoptic_seg is not None), fx, fy, cx, cy, h, len(ids),
                          int(road_class_id if road_class_id is not None else -1))
    for i, v in enumerate(ids):
        cfg.filter_ids[i] = int(v)
    pan = None if panoptic_seg is None else panoptic_seg.reshape(H, W).long().contiguous()
    depth, xyz, _ = _C.depth_post(cfg, depth_logits.reshape(H, W).float().contiguous(), pan)
    return depth, xyz
