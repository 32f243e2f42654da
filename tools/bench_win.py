"""Windowed 3x3 kernel (csrc/conv_win.hip) vs the generic implicit-GEMM kernels: parity on ragged shapes, then time per layer
shape of the benchmark step.  `python tools/bench_win.py [--check-only]`"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgnet_amd import _C


def cl(*s, dtype=torch.bfloat16):
    return torch.randn(*s, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)


def generic(x, wl, res=None):
    os.environ["MGN_CONV_WIN"] = "0"
    y = _C.conv_igemm(x, wl, x.shape[2:], None, 1, 1, residual=res)
    del os.environ["MGN_CONV_WIN"]
    return y


def check():
    torch.manual_seed(0)
    for dtype in (torch.bfloat16, torch.float16):
        for (N, Cin, Cout, H, W) in [(1, 32, 128, 16, 32), (2, 128, 128, 12, 20), (1, 64, 256, 33, 70), (3, 96, 128, 7, 5),
                                     (2, 256, 256, 19, 37), (1, 512, 128, 9, 40), (2, 128, 384, 40, 64)]:
            x = cl(N, Cin, H, W, dtype=dtype)
            w = torch.randn(Cout, Cin, 3, 3, device="cuda") / (Cin * 9) ** 0.5
            wl = w.permute(0, 2, 3, 1).contiguous().to(dtype)
            res = cl(N, Cout, H, W, dtype=dtype)
            ref = torch.nn.functional.conv2d(x.double(), wl.permute(0, 3, 1, 2).double(), padding=1)
            for pr in (16, 8):
                for r in (None, res):
                    y = _C.conv3x3_win(x, wl, residual=r, patch_rows=pr)
                    rr = ref if r is None else ref + r.double()
                    err = float((y.double() - rr).abs().max() / rr.abs().max())
                    g = generic(x, wl, r)
                    dg = float((y.float() - g.float()).abs().max() / g.float().abs().max())
                    ok = err < 6e-3 and dg < 8e-3
                    print(f"{str(dtype)[6:]:9s} N{N} {Cin}->{Cout} {H}x{W} patch{pr} res={r is not None}: vs fp64 {err:.2e}  vs generic {dg:.2e} "
                          f"{'OK' if ok else 'FAIL'}", flush=True)
                    assert ok


def timeit(f, n=20):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def bench():
    B = 8
    for (Cin, Cout, H, W) in [(128, 128, 128, 256), (256, 256, 128, 256), (256, 256, 64, 128), (512, 512, 32, 64), (128, 128, 64, 128),
                              (128, 256, 64, 128), (256, 128, 64, 128), (256, 256, 32, 64), (512, 128, 32, 64), (128, 512, 32, 64)]:
        x = cl(B, Cin, H, W)
        wl = (torch.randn(Cout, 3, 3, Cin, device="cuda") / (Cin * 9) ** 0.5).to(torch.bfloat16)
        fl = 2.0 * B * H * W * Cin * Cout * 9
        line = f"{Cin}->{Cout} @{H}x{W}:"
        res = cl(B, Cout, H, W)
        for pr in (16, 8):
            t = timeit(lambda: _C.conv3x3_win(x, wl, patch_rows=pr))
            tr = timeit(lambda: _C.conv3x3_win(x, wl, residual=res, patch_rows=pr))
            line += f" | win{pr} {t:7.1f} us ({fl / t / 1e6:6.0f} TF/s), +residual {tr:7.1f} us"
        print(line, flush=True)


if __name__ == "__main__":
    check()
    if "--check-only" not in sys.argv:
        bench()
