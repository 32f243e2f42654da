"""Times the device-side panoptic target generation (mgn_panoptic_targets) on Cityscapes-shaped label images and prints
one JSON line with its HBM roofline figure, next to the numpy oracle timed on the host cores.

    python tools/bench_targets.py [--batch 8] [--height 1024] [--width 2048] [--iters 50] [--rgb]
"""
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
from mgnet_amd.data import PanopticDeepLabTargetGenerator  # noqa: E402


def synth():
    spec = importlib.util.spec_from_file_location("mk", os.path.join(ROOT, "tests", "golden", "make_golden_targets.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--rgb", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    mk = synth()
    B, H, W = a.batch, a.height, a.width
    cases = [mk.synth_case(seed=100 + b, H=H, W=W, n_stuff=9, n_things=60, small_blobs=25, n_crowd=4, n_absent=3) for b in range(B)]
    pan = np.stack([c[0] for c in cases])
    segs = [c[1] for c in cases]
    kw = dict(ignore_label=255, thing_ids=mk.THING_IDS, sigma=8, small_instance_area=4096, small_instance_weight=3)
    g = PanopticDeepLabTargetGenerator(depth_ignore_ids=[10], **kw)
    if a.rgb:
        pan_in = np.stack([pan & 255, (pan >> 8) & 255, (pan >> 16) & 255], -1).astype(np.uint8)
    else:
        pan_in = pan
    dev = torch.from_numpy(pan_in).cuda()
    # kernel-only timing: tables already on the device, HIP events around the three launches
    tab, cnt, cap, _ = g._tables(segs)
    cfg = g._cfg(B, H, W, a.rgb, cap)
    gd = torch.from_numpy(g.g.astype(np.float32).reshape(-1)).cuda()
    ids, attr, n = torch.from_numpy(tab[0]).cuda(), torch.from_numpy(tab[1]).cuda(), torch.from_numpy(cnt).cuda()
    for _ in range(5):
        out = _C.panoptic_targets(cfg, dev, ids, attr, n, gd, want_mask=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        out = _C.panoptic_targets(cfg, dev, ids, attr, n, gd, want_mask=True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    # end to end from HOST label images (H2D of the labels + tables included)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.generate_batch(pan_in, segs)
    torch.cuda.synchronize()
    ms_host = (time.perf_counter() - t0) / 10 * 1e3
    label_bytes = 3 if a.rgb else 4
    algo = (2 * label_bytes + 8 + 4 + 8 + 4 + 4 + 4 + 1) * B * H * W
    line = {"metric": "panoptic_target_generation", "value": B / ms * 1e3, "unit": "frames/s", "ms_per_batch": ms,
            "ms_per_batch_from_host_labels": ms_host, "dtype": "u8/int32/f32/f64",
            "config": {"workload": f"{B} label images {H}x{W}, ~{len(segs[0])} segments each", "labels": "rgb" if a.rgb else "int32"},
            "roofline": {"bound": "hbm", "achieved": algo / ms / 1e6, "peak": 8000.0, "unit": "GB/s", "frac": algo / ms / 1e6 / 8000.0,
                         "algorithmic_bytes": algo, "traffic": None}}
    if not a.no_cpu:
        sys.path.insert(0, ROOT)
        from oracle import target_oracle as TO
        t0 = time.perf_counter()
        ref = TO.panoptic_targets(pan[0], segs[0], depth_ignore_ids=[10], **kw)
        t_cpu = time.perf_counter() - t0
        ok = all(np.array_equal(out[k][0].cpu().numpy(), np.asarray(ref[k])) for k in ("sem_seg", "center", "offset", "sem_seg_weights", "center_weights", "offset_weights"))
        line["cpu_baseline"] = {"value": 1.0 / t_cpu, "unit": "frames/s", "cores": 1, "kind": "port", "sample": "1 frame, numpy oracle", "bit_exact_vs_gpu": bool(ok)}
    print(json.dumps(line))


if __name__ == "__main__":
    main()
