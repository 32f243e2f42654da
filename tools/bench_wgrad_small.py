#!/usr/bin/env python3
"""Weight-gradient launches of the 1x1 and strided layers of the C4 step (the split-K tile kernels `conv_wgrad` / `conv_wgrad_tr` + reduce):
µs per shape, checked against the fp32 torch gradient.  `MGN_WGRAD_MINPX=2048` restores the round-3 split floor for an A/B on one box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mgnet_amd import _C

dev = torch.device("cuda:0")
B = 8
# Cin, Cout, IH, IW, k, stride, calls per step
SHAPES = [(256, 256, 128, 256, 1, 1, 3), (256, 32, 128, 256, 1, 1, 4), (32, 256, 128, 256, 1, 1, 4), (256, 512, 64, 128, 1, 2, 2),
          (128, 256, 128, 256, 1, 2, 2), (64, 128, 256, 512, 1, 2, 2), (256, 32, 32, 64, 1, 1, 2), (32, 256, 32, 64, 1, 1, 2),
          (512, 256, 32, 64, 1, 1, 1), (256, 32, 64, 128, 1, 1, 1), (256, 512, 32, 64, 1, 1, 1), (64, 128, 256, 512, 3, 2, 2),
          (256, 512, 64, 128, 3, 2, 2), (128, 256, 128, 256, 3, 2, 2), (512, 128, 32, 64, 3, 1, 3), (256, 256, 32, 64, 3, 1, 2), (512, 512, 32, 64, 3, 1, 6),
          (128, 128, 64, 128, 3, 1, 3), (256, 128, 64, 128, 3, 1, 3), (256, 256, 64, 128, 3, 1, 6)]


def cl(*shape):
    return torch.randn(*shape, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)


tot = 0.0
for Cin, Cout, IH, IW, k, s, cnt in SHAPES:
    pad = k // 2
    OH, OW = (IH + 2 * pad - k) // s + 1, (IW + 2 * pad - k) // s + 1
    x, dy = cl(B, Cin, IH, IW), cl(B, Cout, OH, OW)
    f = lambda: _C.conv_wgrad(dy, x, k, k, s, pad)
    got = f()
    ref = torch.nn.grad.conv2d_weight(x.float(), (Cout, Cin, k, k), dy.float(), stride=s, padding=pad)
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err < 2e-3, (Cin, Cout, err)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    n = 30
    t = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t) / n * 1e6
    mb = (x.numel() + dy.numel()) * 2 / 1e6
    tot += cnt * us
    print(f"wgrad {k}x{k} s{s} {Cin:3d}->{Cout:3d} @{IH}x{IW}: {us:7.1f} us  {mb / us:6.2f} TB/s of inputs  x{cnt}  err {err:.1e}")
print(f"sum over the step: {tot / 1e3:.3f} ms")
