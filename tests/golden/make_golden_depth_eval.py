"""Generates tests/golden/depth_eval.npz from the reference's OWN DepthEvaluator.process
(mgnet/evaluation/depth_evaluation.py, imported unmodified).  detectron2 is absent, so three names it imports are supplied
as stand-ins: `detection_utils.read_image` (returns the arrays of this script by "file name"), `CityscapesEvaluator` (a base
class that only provides `_cpu_device`) and `comm` (unused by `process`).  Runs only where /root/reference exists:

    python tests/golden/make_golden_depth_eval.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/mgnet/evaluation/depth_evaluation.py"


def depth_eval_case(seed, H, W):
    """Prediction = ground truth x smooth error field x a global scale error; 30 % of the ground truth missing (0), some
    beyond max_depth, some predictions outside [min, max]."""
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    gt = (4.0 + 70.0 * (1.0 - yy / H) ** 2 + 3.0 * np.sin(xx / 17.0)).astype(np.float32)
    gt[rs.rand(H, W) < 0.3] = 0.0
    gt[rs.rand(H, W) < 0.02] = 95.0
    pred = (np.abs(gt) + 5.0) * (1.0 + 0.15 * np.sin(xx / 9.0 + yy / 13.0)).astype(np.float32) * 0.8
    pred = pred + rs.randn(H, W).astype(np.float32) * 0.3
    pred[rs.rand(H, W) < 0.01] = 200.0
    pred[rs.rand(H, W) < 0.01] = 1e-5
    return pred.astype(np.float32), gt.astype(np.float32)


CASES = {
    "plain": dict(seed=1, H=48, W=80, use_gt_scale=False, use_eigen_crop=False, disparity=False),
    "gt_scale": dict(seed=2, H=48, W=80, use_gt_scale=True, use_eigen_crop=False, disparity=False),
    "eigen_crop_scale": dict(seed=3, H=37, W=121, use_gt_scale=True, use_eigen_crop=True, disparity=False),
    "disparity": dict(seed=4, H=40, W=64, use_gt_scale=False, use_eigen_crop=True, disparity=True),
}
CALIB = {"extrinsic": {"baseline": 0.22}, "intrinsic": {"fx": 2262.52}}


def label_file(gt, disparity):
    """The array the reference's reader would return for this ground truth."""
    if not disparity:
        return np.round(gt * 256.0).astype(np.uint16)            # KITTI: 16-bit PNG, depth * 256
    d = np.zeros_like(gt)
    m = gt > 0
    d[m] = CALIB["extrinsic"]["baseline"] * CALIB["intrinsic"]["fx"] / gt[m]
    return np.where(m, np.round(d * 256.0 + 1.0), 0).astype(np.uint16)   # Cityscapes disparity encoding


def load_reference(files):
    def mod(name, **attrs):
        m = sys.modules.get(name) or types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class CityscapesEvaluator:
        def __init__(self, dataset_name):
            self._cpu_device = torch.device("cpu")
    for n in ("detectron2", "detectron2.data", "detectron2.evaluation", "detectron2.utils"):
        mod(n)
    mod("detectron2.data.detection_utils", read_image=lambda name, format=None: files[name].copy())
    sys.modules["detectron2.data"].detection_utils = sys.modules["detectron2.data.detection_utils"]
    mod("detectron2.evaluation.cityscapes_evaluation", CityscapesEvaluator=CityscapesEvaluator)
    mod("detectron2.utils.comm")
    sys.modules["detectron2.utils"].comm = sys.modules["detectron2.utils.comm"]
    spec = importlib.util.spec_from_file_location("ref_depth_evaluation", REF)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.DepthEvaluator


def main():
    files, out = {}, {}
    Ev = load_reference(files)
    for name, c in CASES.items():
        pred, gt = depth_eval_case(c["seed"], c["H"], c["W"])
        files[name] = label_file(gt, c["disparity"])
        ev = Ev("x", use_gt_scale=c["use_gt_scale"], use_eigen_crop=c["use_eigen_crop"])
        inp = {("disparity_file_name" if c["disparity"] else "depth_file_name"): name, "calibration_info": CALIB}
        ev.process([inp], [{"depth": (torch.from_numpy(pred.copy()), None)}])
        out[name + ".errors"] = np.array(ev._errors[0], dtype=np.float64)
        out[name + ".ratio"] = np.array(ev._ratios[0] if ev._ratios else 1.0, dtype=np.float64)
        out[name + ".file"] = files[name]
        print(name, out[name + ".errors"].round(5), float(out[name + ".ratio"]))
    np.savez_compressed(os.path.join(HERE, "depth_eval.npz"), **out)
    print(os.path.getsize(os.path.join(HERE, "depth_eval.npz")), "bytes")


if __name__ == "__main__":
    if not os.path.exists(REF):
        sys.exit("reference not present: fixtures can only be regenerated in the build container")
    main()
