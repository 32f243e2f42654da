#!/usr/bin/env python3
"""conv3x3_c64 (64-channel 3x3: ResNet layer1 forward / data gradient) at the C4 shape: plain, with the fused skip gradient (residual),
and through ops.conv2d with the statistics epilogue; checked against torch's fp32 convolution of the same bf16 operands."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mgnet_amd import _C

dev = torch.device("cuda:0")
B, H, W = int(os.environ.get("B", 8)), 256, 512


def cl(*shape):
    return torch.randn(*shape, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


x, res = cl(B, 64, H, W), cl(B, 64, H, W)
w = torch.nn.Parameter(torch.randn(64, 64, 3, 3, device=dev) / 24.0)
wl = _C.weight_layout(w, 0)
ref = torch.nn.functional.conv2d(x[:1].float(), w.detach().to(torch.bfloat16).float(), padding=1)
y = _C.conv_igemm(x, wl, (H, W), None, 1, 1)
err = float((y[:1].float() - ref).abs().max() / ref.abs().max())
y2 = _C.conv_igemm(x, wl, (H, W), None, 1, 1, residual=res)
err2 = float((y2[:1].float() - (ref + res[:1].float())).abs().max() / ref.abs().max())
assert err < 8e-3 and err2 < 1.2e-2, (err, err2)
gf = 2.0 * B * H * W * 64 * 64 * 9 / 1e9
for name, f in (("plain", lambda: _C.conv_igemm(x, wl, (H, W), None, 1, 1)), ("residual", lambda: _C.conv_igemm(x, wl, (H, W), None, 1, 1, residual=res))):
    us = timeit(f)
    print(f"conv3x3_c64 64->64 @{H}x{W} {name:9s}: {us:7.1f} us  {gf / us * 1e3:6.0f} TF/s  err {err:.1e}/{err2:.1e}  lib={os.environ.get('MGNET_HIP_LIB', 'in-tree')}")
