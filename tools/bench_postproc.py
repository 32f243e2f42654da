"""Times the panoptic fusion (mgn_panoptic_post) on a Cityscapes-sized frame; prints one JSON line.
    python tools/bench_postproc.py [--height 1024 --width 2048 --instances 100 --iters 50]"""
import argparse
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--instances", type=int, default=100)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    spec = importlib.util.spec_from_file_location("mk", os.path.join(ROOT, "tests", "golden", "make_golden_postproc.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    H, W = a.height, a.width
    sem, center, off = mk.pan_case(seed=7, H=H, W=W, n_inst=a.instances, noise=2.0, rmax=45)
    cfg = _C.PanopticCfg(H, W, 8, 10, 1000, 2048, -1, 0.3, 7)
    s, c, o = torch.from_numpy(sem).cuda(), torch.from_numpy(center).cuda(), torch.from_numpy(off).cuda()
    for _ in range(3):
        pan, info = _C.panoptic_post(cfg, s, c, o)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        pan, info = _C.panoptic_post(cfg, s, c, o)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    algo = (8 + 4 + 8 + 8) * H * W   # sem int64 + heat map + offsets read, panoptic int64 written
    line = {"metric": "panoptic_fusion", "value": 1e3 / ms, "unit": "frames/s", "ms_per_frame": ms, "dtype": "int64/f32",
            "config": {"workload": f"1 frame {H}x{W}", "centres": int(info[0]), "thing_px": int((sem > 10).sum())},
            "roofline": {"bound": "hbm", "achieved": algo / ms / 1e6, "peak": 8000.0, "unit": "GB/s", "frac": algo / ms / 1e6 / 8000.0,
                         "algorithmic_bytes": algo, "traffic": None}}
    if not a.no_cpu:
        from oracle import postproc_oracle as PO
        t0 = time.perf_counter()
        ref = PO.panoptic_prediction(sem, center, off, **mk.PAN_KW, stuff_area=2048, threshold=0.3, nms_kernel=7)
        t = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": 1.0 / t, "unit": "frames/s", "cores": 1, "kind": "port", "sample": "1 frame, numpy oracle",
                                "bit_exact_vs_gpu": bool(np.array_equal(ref, pan.cpu().numpy()))}
    print(json.dumps(line))


if __name__ == "__main__":
    main()
