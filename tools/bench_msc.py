#!/usr/bin/env python3
"""Throughput of the multi-scale + flip accumulation kernels (csrc/mscflip.hip) at the Cityscapes test size: one 1024 x 2048
frame, stride-8 head outputs of the seven scales, 20 + 1 + 2 + 1 channels; one JSON line."""
import json, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mgnet_amd import _C

H, W, stride = 1024, 2048, 8
scales = [0.5, 0.75, 1.0, 1.25, 1.5, 1.75, 2.0]
dev = "cuda"
norm = torch.randn(1, 3, H, W, device=dev)
heads = [("softmax", 20), ("plain", 1), ("offset", 2), ("inv2depth", 1)]
acc = {m: torch.empty(1, c, H, W, device=dev) for m, c in heads}
lrs = {}
for s in scales:
    h, w = int(math.floor(H * s)) // stride, int(math.floor(W * s)) // stride
    lrs[s] = {m: (torch.rand(1, 32, h, w, device=dev) + 0.1).bfloat16().contiguous(memory_format=torch.channels_last)[:, :c] for m, c in heads}


def passes():
    k, n = 0, 2 * len(scales)
    for s in scales:
        for f in (0, 1):
            _C.msc_input(norm, int(math.floor(H * s)), int(math.floor(W * s)), f, torch.bfloat16)
            for m, c in heads:
                _C.msc_accumulate(acc[m], lrs[s][m], m, f, k == 0, stride=float(stride), scale=float(s), divide=float(n) if k == n - 1 else 0.0)
            k += 1


for _ in range(3):
    passes()
torch.cuda.synchronize()
t0 = time.perf_counter()
n_it = 10
for _ in range(n_it):
    passes()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n_it
px = H * W
acc_bytes = 14 * 24 * px * 8 - 24 * px * 4          # read-modify-write of 24 fp32 channels per pass (the first pass only writes)
in_bytes = sum(int(math.floor(H * s)) * int(math.floor(W * s)) * (16 + 12 * 1.0) for s in scales) * 2   # 16 B written per pixel, ~12 B read
print(json.dumps({"what": "multi-scale + flip accumulation, 14 passes, one 1024x2048 frame (inputs + 4 head outputs per pass)",
                  "ms_per_frame": round(dt * 1e3, 3), "algorithmic_GB": round((acc_bytes + in_bytes) / 1e9, 3),
                  "GB_per_s": round((acc_bytes + in_bytes) / dt / 1e9, 1), "frac_of_8TBs": round((acc_bytes + in_bytes) / dt / 8e12, 3)}))
