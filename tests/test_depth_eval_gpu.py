"""Depth metrics on the device (mgn_depth_metrics behind mgnet_amd.evaluation.DepthEvaluator) against the reference's own
DepthEvaluator.process (tests/golden/depth_eval.npz, tests/golden/make_golden_depth_eval.py).  The reference accumulates
in float32 (numpy pairwise sums), the kernel in float64: agreement to ~1e-6 relative."""
import os

import numpy as np
import pytest
import torch
from conftest import GOLDEN, golden_script

pytestmark = pytest.mark.gpu
MK = golden_script("make_golden_depth_eval")
Z = np.load(os.path.join(GOLDEN, "depth_eval.npz"))


@pytest.mark.parametrize("name", list(MK.CASES))
def test_metrics_match_reference(name):
    from mgnet_amd.evaluation import DepthEvaluator
    c = MK.CASES[name]
    pred, _ = MK.depth_eval_case(c["seed"], c["H"], c["W"])
    files = {name: Z[name + ".file"]}
    ev = DepthEvaluator("x", use_gt_scale=c["use_gt_scale"], use_eigen_crop=c["use_eigen_crop"], read_image=lambda n: files[n].copy())
    inp = {("disparity_file_name" if c["disparity"] else "depth_file_name"): name, "calibration_info": MK.CALIB}
    ev.process([inp], [{"depth": (torch.from_numpy(pred).cuda(), None)}])
    res = ev.evaluate()["depth"]
    want = Z[name + ".errors"]
    got = np.array(list(res.values()))
    np.testing.assert_allclose(got, want, rtol=5e-6, atol=1e-9)
    if c["use_gt_scale"]:
        assert ev.scale_ratio_median == pytest.approx(float(Z[name + ".ratio"]), rel=1e-6)
    assert list(res.keys()) == ["Abs Rel", "Sq Rel", "RMSE", "RMSE log", "δ < 1.25", "δ < 1.25²", "δ < 1.25³"]


def test_full_frame_and_mean_over_frames():
    from mgnet_amd.evaluation import DepthEvaluator
    ev = DepthEvaluator("x", use_gt_scale=True)
    rows = []
    for seed in (7, 8):
        pred, gt = MK.depth_eval_case(seed, 375, 1242)
        ev.process([{"depth": gt}], [{"depth": (torch.from_numpy(pred).cuda(), None)}])
        # numpy restatement in float64 of depth_evaluation.py:70-110
        m = (gt > 0.001) & (gt < 80.0)
        p, l = pred[m].astype(np.float32), gt[m]
        p = p * (np.median(l) / np.median(p))
        p = np.clip(p, 0.001, 80.0)
        th = np.maximum(l / p, p / l)
        d = (l - p).astype(np.float64)
        rows.append([np.mean(np.abs(d) / l), np.mean(d ** 2 / l), np.sqrt(np.mean(d ** 2)),
                     np.sqrt(np.mean((np.log(l).astype(np.float64) - np.log(p)) ** 2)), (th < 1.25).mean(), (th < 1.25 ** 2).mean(),
                     (th < 1.25 ** 3).mean()])
    got = np.array(list(ev.evaluate()["depth"].values()))
    np.testing.assert_allclose(got, np.mean(rows, 0), rtol=2e-5)
    ev.reset()
    assert ev._errors == []
    with pytest.raises(RuntimeError):
        ev.process([{"depth": gt}], [{"depth": (torch.from_numpy(pred), None)}])
    with pytest.raises(RuntimeError):
        DepthEvaluator("x").process([{}], [{"depth": (torch.from_numpy(pred).cuda(), None)}])
