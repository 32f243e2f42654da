#!/usr/bin/env python3
"""Standalone bandwidth of the IABN kernels (same tensor re-read in a loop: cache-warm upper bound) on MGNet layer shapes."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgnet_amd import _C


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


for shape in [(8, 64, 512, 1024), (8, 64, 256, 512), (8, 128, 128, 256), (8, 256, 128, 256), (8, 256, 64, 128), (8, 512, 32, 64)]:
    N, C, H, W = shape
    x = torch.randn(*shape, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn_like(x)
    dx = torch.empty_like(x)
    w, b = torch.rand(C, device="cuda") + 0.5, torch.zeros(C, device="cuda")
    M = N * H * W
    T = x.numel() * 2
    coef = _C.iabn_train_coeffs(x, M, C, w, b, 1e-5, 0.01, None, None)
    y = x.clone()
    _C.iabn_apply(x, y, M, C, coef[0], coef[1], 1, 0.01)
    t_s = timeit(lambda: _C.iabn_train_coeffs(x, M, C, w, b, 1e-5, 0.01, None, None))
    t_a = timeit(lambda: _C.iabn_apply(x, y, M, C, coef[0], coef[1], 1, 0.01))
    t_r = timeit(lambda: _C.iabn_bwd_reduce(y, dy, M, C, w, b, 1e-5, 1, 0.01))
    sums, _, _ = _C.iabn_bwd_reduce(y, dy, M, C, w, b, 1e-5, 1, 0.01)
    t_b = timeit(lambda: _C.iabn_bwd_apply(y, dy, dx, M, C, w, b, coef[2:], sums, float(M), 1e-5, 1, 0.01))
    print(f"{str(shape):24s} {T/1e6:6.0f} MB | stats {t_s*1e6:6.1f} us {T/t_s/1e12:5.2f} TB/s | apply {t_a*1e6:6.1f} us {2*T/t_a/1e12:5.2f} | "
          f"bwd_reduce {t_r*1e6:6.1f} us {2*T/t_r/1e12:5.2f} | bwd_apply {t_b*1e6:6.1f} us {3*T/t_b/1e12:5.2f}", flush=True)
