"""Depth metrics -- host-side mirror of mgnet/evaluation/depth_evaluation.py (DepthEvaluator: same constructor arguments,
`reset / process / evaluate`, same result keys) with the per-frame arithmetic on the device (`mgn_depth_metrics`).

Differences kept deliberately small: the ground truth comes from the input dict as an array/tensor under "depth" (metric
depth) or from the reference's own file keys "depth_file_name" (16-bit PNG / 256) / "disparity_file_name" (Cityscapes
disparity + calibration) when a `read_image` callable is supplied -- detectron2's file reader is not part of this
package; multi-process gathering (detectron2 `comm`) is replaced by torch.distributed.all_gather_object."""
from collections import OrderedDict

import numpy as np
import torch

from .. import _C

__all__ = ["DepthEvaluator"]

NAMES = ["Abs Rel", "Sq Rel", "RMSE", "RMSE log", "δ < 1.25", "δ < 1.25²", "δ < 1.25³"]


class DepthEvaluator:
    def __init__(self, dataset_name=None, min_depth=0.001, max_depth=80.0, use_gt_scale=False, use_eigen_crop=False,
                 read_image=None):
        self._dataset_name = dataset_name
        self._min_depth, self._max_depth = min_depth, max_depth
        self._use_gt_scale, self._use_eigen_crop = use_gt_scale, use_eigen_crop
        self._read_image = read_image
        self.reset()

    def reset(self):
        self._errors, self._ratios = [], []

    def _label(self, input_):
        if "depth" in input_:
            return np.asarray(input_["depth"].cpu() if torch.is_tensor(input_["depth"]) else input_["depth"], dtype=np.float32)
        if self._read_image is None:
            raise RuntimeError("ground truth: pass the metric depth as input['depth'] or give DepthEvaluator a read_image callable")
        if "depth_file_name" in input_:   # depth_evaluation.py:54-55
            return self._read_image(input_["depth_file_name"]).astype(np.float32) / 256.0
        if "disparity_file_name" in input_:   # :56-65
            label = self._read_image(input_["disparity_file_name"]).astype(np.float32)
            label[label != 0] = (label[label != 0] - 1.0) / 256.0
            factor = input_["calibration_info"]["extrinsic"]["baseline"] * input_["calibration_info"]["intrinsic"]["fx"]
            label[label != 0] = factor / label[label != 0]
            return label
        raise RuntimeError("Neither depth_file_name nor disparity_file_name are given for the dataset. "
                           "Impossible to run DepthEvaluator!")

    def process(self, inputs, outputs):
        for input_, output in zip(inputs, outputs):
            pred = output["depth"][0]
            if not pred.is_cuda:
                raise RuntimeError("DepthEvaluator runs on the GPU (no CPU fallback by design)")
            label = self._label(input_)
            H, W = label.shape[-2:]
            crop = (0, H, 0, W)
            if self._use_eigen_crop:   # :74-85 (float64 products truncated to int32, as numpy does)
                c = np.array([0.40810811 * H, 0.99189189 * H, 0.03594771 * W, 0.96405229 * W]).astype(np.int32)
                crop = tuple(int(v) for v in c)
            out = _C.depth_metrics(pred.reshape(H, W).float().contiguous(),
                                   torch.from_numpy(np.ascontiguousarray(label)).to(pred.device), self._min_depth,
                                   self._max_depth, self._use_gt_scale, crop)
            self._errors.append(out)   # stays on the device until evaluate(): no sync per frame

    def evaluate(self):
        errs = torch.stack(self._errors).cpu().numpy() if self._errors else np.zeros((0, 9))
        rows, ratios = [list(e[:7]) for e in errs], [float(e[7]) for e in errs] if self._use_gt_scale else []
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            gathered = [None] * torch.distributed.get_world_size()
            torch.distributed.all_gather_object(gathered, (rows, ratios))
            rows = [r for g in gathered for r in g[0]]
            ratios = [r for g in gathered for r in g[1]]
            if torch.distributed.get_rank() != 0:
                return None
        mean_errors = np.array(rows).mean(0)
        ret = OrderedDict()
        ret["depth"] = {k: float(v) for k, v in zip(NAMES, mean_errors)}
        if self._use_gt_scale:
            med = float(np.median(ratios))
            self.scale_ratio_median, self.scale_ratio_std = med, float(np.std(np.array(ratios) / med))
        return ret
