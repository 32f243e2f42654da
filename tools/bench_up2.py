#!/usr/bin/env python3
"""Data gradient of the 3x3 / stride-2 layers of the C4 step: csrc/conv_up2.hip (one low-resolution window, four parity classes) vs the
parity-class launch of the generic implicit GEMM (MGN_CONV_NOUP2WIN=1), µs per call on one box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mgnet_amd import _C

dev = torch.device("cuda:0")
SHAPES = [(64, 128, 256, 512), (128, 256, 128, 256), (256, 512, 64, 128)]   # Cin, Cout of the forward conv, input H, W


def run(f, n=30):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


for ks, res in ((3, False), (3, True), (1, False)):
    for Cin, Cout, H, W in SHAPES:
        w = torch.randn(Cout, Cin, ks, ks, device=dev) / (Cin * ks * ks) ** 0.5
        dy = torch.randn(8, Cout, H // 2, W // 2, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        r = torch.randn(8, Cin, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) if res else None
        wl = _C._weight_layout_now(w, 1, 0, None, 0, torch.bfloat16)
        f = lambda: _C.conv_igemm(dy, wl, (H, W), None, 1, ks // 2, up=2, residual=r)
        os.environ.pop("MGN_CONV_NOUP2WIN", None)
        new = run(f)
        os.environ["MGN_CONV_NOUP2WIN"] = "1"
        old = run(f)
        os.environ.pop("MGN_CONV_NOUP2WIN", None)
        gf = 2.0 * 8 * (H // 2) * (W // 2) * Cin * Cout * ks * ks / 1e9
        mb = (dy.numel() + 8 * Cin * H * W * (2 if res else 1)) * 2 / 1e6
        print(f"dgrad {ks}x{ks} s2 {Cout:3d}->{Cin:3d} out {H}x{W} residual={int(res)}: window {new:6.1f} us ({gf / new * 1e3:5.0f} TF/s, {mb / new:5.2f} TB/s)   "
              f"classes {old:6.1f} us")
