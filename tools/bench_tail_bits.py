"""Block-tail ReLU mask: the backward reduction reading the mask bytes (abn_add_relu_fwd want_bits) against reading the output map, and what
the forward pays for writing them -- isolated, at the benchmark's four block-tail shapes.  python tools/bench_tail_bits.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mgnet_amd import _C  # noqa: E402


def timed(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    tot = [0.0] * 4
    for shape in [(8, 64, 256, 512), (8, 128, 128, 256), (8, 256, 64, 128), (8, 512, 32, 64)]:
        N, C, H, W = shape
        M = N * H * W
        x = torch.randn(shape, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        sc = torch.randn_like(x)
        g = torch.randn_like(x)
        coef = torch.stack([torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1, torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")]).contiguous()
        w32, b32 = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
        y, bits = _C.abn_add_relu_fwd(x, coef, sc, want_bits=True)
        t = [timed(lambda: _C.abn_add_relu_fwd(x, coef, sc)), timed(lambda: _C.abn_add_relu_fwd(x, coef, sc, want_bits=True)),
             timed(lambda: _C.iabn_bwd_reduce_x_relu(x, g, y, M, C, w32, b32, coef, 1e-5)),
             timed(lambda: _C.iabn_bwd_reduce_x_relu(x, g, None, M, C, w32, b32, coef, 1e-5, relu_bits=bits))]
        mb = x.numel() * 2 / 1e6
        print(f"{shape}  {mb:6.1f} MB | fwd {t[0]:7.1f} us -> with bits {t[1]:7.1f} | bwd reduce from y {t[2]:7.1f} us ({4 * mb / t[2]:.2f} TB/s) -> from bits {t[3]:7.1f} us ({3.0625 * mb / t[3]:.2f} TB/s)")
        tot = [a + b for a, b in zip(tot, t)]
    print(f"per step (4 shapes x 2 blocks x 2 networks): fwd {4 * tot[0] / 1e3:.3f} -> {4 * tot[1] / 1e3:.3f} ms, bwd reduce {4 * tot[2] / 1e3:.3f} -> {4 * tot[3] / 1e3:.3f} ms")


if __name__ == "__main__":
    main()
