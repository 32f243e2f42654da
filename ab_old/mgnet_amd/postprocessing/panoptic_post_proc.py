"""Panoptic fusion -- host-side mirror of mgnet/postprocessing/panoptic_post_proc.py:9-71 (same function name, arguments,
argument checks and result) over `mgn_panoptic_post` (csrc/postproc.hip)."""
import torch

from .. import _C

__all__ = ["get_panoptic_prediction"]


def get_panoptic_prediction(sem_seg, center_heatmap, offsets, num_thing_classes, last_stuff_id, label_divisor, stuff_area,
                            void_label, threshold=0.3, nms_kernel=7, check=True):
    """sem_seg [1,H,W] predicted labels, center_heatmap [1,H,W], offsets [2,H,W] (dy, dx) -> panoptic [H,W] int64.
    Unlike the reference the inputs are left untouched (it adds the pixel grid into `offsets` and scatters the instance
    ids into `sem_seg`).  `check=True` reads back one flag (a host sync, as the reference's `.item()` at :136) and raises
    if more than 65534 centres survived the NMS -- the point where the reference's own 65535 sentinel (:112-126) breaks."""
    if sem_seg.dim() != 3 and sem_seg.size(0) != 1:   # the reference's checks (:43-50), verbatim semantics
        raise ValueError("Semantic prediction with un-supported shape: {}.".format(sem_seg.size()))
    if center_heatmap.dim() != 3:
        raise ValueError("Center prediction with un-supported dimension: {}.".format(center_heatmap.dim()))
    if offsets.dim() != 3:
        raise ValueError("Offset prediction with un-supported dimension: {}.".format(offsets.dim()))
    if not sem_seg.is_cuda:
        raise RuntimeError("get_panoptic_prediction runs on the GPU (no CPU fallback by design)")
    H, W = sem_seg.shape[-2:]
    cfg = _C.PanopticCfg(H, W, int(num_thing_classes), int(last_stuff_id), int(label_divisor), int(stuff_area),
                         int(void_label), float(threshold), int(nms_kernel))
    pan, info = _C.panoptic_post(cfg, sem_seg.reshape(H, W).long().contiguous(),
                                 center_heatmap.reshape(H, W).float().contiguous(),   # custom_fwd(cast_inputs=float32) (:9)
                                 offsets.float().contiguous())
    if check and int(info[1]):
        raise RuntimeError(f"{int(info[0])} centre points after NMS (limit {_C.PANOPTIC_MAX_CENTERS})")
    return pan
