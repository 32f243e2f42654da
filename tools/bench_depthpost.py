"""Times the depth post-processing (mgn_depth_post: DGC rescaling) on a Cityscapes-sized frame; prints one JSON line."""
import importlib.util
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mgnet_amd import _C  # noqa: E402


def main():
    spec = importlib.util.spec_from_file_location("mk", os.path.join(ROOT, "tests", "golden", "make_golden_postproc.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    H, W = 1024, 2048
    depth, K, pan = mk.depth_case(3, H, W)
    cfg = _C.DepthPostCfg(H, W, 1, 1, float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), 1.65, 2, 0)
    cfg.filter_ids[0], cfg.filter_ids[1] = 10000, 2000
    d, p = torch.from_numpy(depth).cuda(), torch.from_numpy(pan).cuda()
    for _ in range(3):
        out = _C.depth_post(cfg, d, p)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        out = _C.depth_post(cfg, d, p)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    algo = (4 + 8 + 4 + 12) * H * W   # depth + panoptic read, depth + points written (the select's key passes are not algorithmic)
    line = {"metric": "depth_post_processing", "value": 1e3 / ms, "unit": "frames/s", "ms_per_frame": ms, "dtype": "f32",
            "config": {"workload": f"1 frame {H}x{W}, DGC with panoptic road mask"}, "scale": float(out[2]),
            "roofline": {"bound": "hbm", "achieved": algo / ms / 1e6, "peak": 8000.0, "unit": "GB/s", "frac": algo / ms / 1e6 / 8000.0,
                         "algorithmic_bytes": algo, "traffic": None}}
    if "--no-cpu" not in sys.argv:
        from oracle import postproc_oracle as PO
        t0 = time.perf_counter()
        _, _, sc = PO.depth_prediction(depth, True, K=K, real_camera_height=1.65, panoptic=pan, **mk.DEPTH_KW)
        line["cpu_baseline"] = {"value": 1.0 / (time.perf_counter() - t0), "unit": "frames/s", "cores": 1, "kind": "port",
                                "sample": "1 frame, numpy oracle", "scale": float(sc)}
    print(json.dumps(line))


if __name__ == "__main__":
    main()
