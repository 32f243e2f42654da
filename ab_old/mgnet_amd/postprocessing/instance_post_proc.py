"""mgnet.postprocessing.get_instance_predictions (instance_post_proc.py:11-72) on the device: csrc/instances.hip through the
C-ABI (`mgn_instance_post`, `mgn_instance_masks`).  Same arguments and result -- a list with one `Instances` per thing segment in
ascending panoptic-id order, fields `pred_classes`, `pred_masks`, `scores`, `pred_boxes` -- computed in one pass over the pixels
instead of one host loop iteration per segment."""
from .. import _C
from ..structures import Boxes, Instances

__all__ = ["get_instance_predictions"]


def get_instance_predictions(sem_seg, center_heatmap, panoptic_image, thing_ids, label_divisor):
    """sem_seg [C,H,W] semantic logits, center_heatmap [1,H,W], panoptic_image [H,W] -> List[Instances]"""
    if not sem_seg.is_cuda:
        raise RuntimeError("get_instance_predictions runs on the GPU (no CPU fallback by design)")
    C, H, W = sem_seg.shape
    labels, classes, scores, boxes, masks = _C.instance_post(sem_seg.float().contiguous(), center_heatmap.reshape(H, W).float().contiguous(),
                                                             panoptic_image.long().contiguous(), list(thing_ids), label_divisor)
    out = []
    for k in range(labels.numel()):   # the reference's list of single-instance structs (views of the batched results)
        out.append(Instances((H, W), pred_classes=classes[k:k + 1], pred_masks=masks[k:k + 1], scores=scores[k:k + 1],
                             pred_boxes=Boxes(boxes[k:k + 1])))
    return out
