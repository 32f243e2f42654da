#!/usr/bin/env python3
"""3x3 / stride-1 weight-gradient kernels (`conv_wgrad3x3_s64/_s128`) at the layer shapes of the C4 step: µs and TFLOP/s per shape
for the library under test (`MGNET_HIP_LIB=<other .so>` for an A/B on the same box), checked against the fp32 torch gradient."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mgnet_amd import _C

dev = torch.device("cuda:0")
B = int(os.environ.get("B", 8))
SHAPES = [(64, 64, 256, 512, 8), (128, 128, 128, 256, 9), (256, 256, 64, 128, 6), (512, 512, 32, 64, 6), (256, 256, 128, 256, 4),
          (512, 128, 32, 64, 3), (256, 128, 64, 128, 3), (128, 256, 64, 128, 1), (128, 256, 32, 64, 1)]   # Cin, Cout, H, W, calls per step


def cl(*shape):
    return torch.randn(*shape, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)


tot = 0.0
for Cin, Cout, H, W, cnt in SHAPES:
    x, dy = cl(B, Cin, H, W), cl(B, Cout, H, W)
    f = lambda: _C.conv_wgrad(dy, x, 3, 3, 1, 1)
    dw = f()
    if os.environ.get("CHECK"):
        ref = torch.nn.grad.conv2d_weight(x[:2].float(), (Cout, Cin, 3, 3), dy[:2].float(), stride=1, padding=1)
        got = _C.conv_wgrad(dy[:2].contiguous(memory_format=torch.channels_last), x[:2].contiguous(memory_format=torch.channels_last), 3, 3, 1, 1)
        err = float((got - ref).abs().max() / ref.abs().max())
        assert err < 2e-3, err
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    n = 20
    t = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t) / n * 1e6
    gf = 2.0 * B * H * W * Cin * Cout * 9 / 1e9
    tot += cnt * us
    print(f"wgrad3x3 {Cin:3d}->{Cout:3d} @{H}x{W}: {us:7.1f} us  {gf / us * 1e-3 * 1e3:6.0f} TF/s  x{cnt}")
print(f"sum over the step's calls: {tot / 1e3:.3f} ms   lib={os.environ.get('MGNET_HIP_LIB', 'in-tree')}")

# the 7x7 / stride-2 stems (conv_wgrad_stem16 / _stem8): channel-padded inputs, cin_real 9 / 3
for Cp, cr in ((16, 9), (8, 3)):
    x, dy = cl(B, Cp, 1024, 2048), cl(B, 64, 512, 1024)
    f = lambda: _C.conv_wgrad(dy, x, 7, 7, 2, 3, cin_real=cr)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t) / 10 * 1e6
    print(f"wgrad stem 7x7 s2 Cp={Cp} (cin {cr}) @1024x2048: {us:7.1f} us")
