"""7x7 stems: persistent windowed kernel (csrc/conv_stem.hip) vs the packed-tap implicit GEMM it replaces -- parity on ragged shapes
(both formats, with statistics), then time at the benchmark shape.  `python tools/bench_stem.py [--check-only]`"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgnet_amd import _C


def run(x, wl, OH, OW, old=False, stats=False):
    if old:
        os.environ["MGN_CONV_NOSTEM7"] = "1"
    holder = []
    y = _C.conv_igemm(x, wl, (OH, OW), None, 2, 3, khw=(7, 7), **(dict(stats=(None, holder)) if stats else {}))
    os.environ.pop("MGN_CONV_NOSTEM7", None)
    return (y, holder[0][0]) if stats else y


def check():
    torch.manual_seed(0)
    for dtype in (torch.bfloat16, torch.float16):
        for (N, Cr, Cp, H, W) in [(2, 3, 8, 40, 56), (1, 9, 16, 34, 46), (3, 3, 8, 17, 130), (2, 9, 16, 64, 64), (1, 3, 8, 7, 9),
                                  (9, 9, 16, 50, 200), (5, 3, 8, 128, 256)]:
            OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
            x = torch.zeros(N, Cp, H, W, device="cuda")
            x[:, :Cr] = torch.randn(N, Cr, H, W, device="cuda") + 0.3
            x = x.to(dtype).contiguous(memory_format=torch.channels_last)
            w = torch.nn.Parameter(torch.randn(64, Cr, 7, 7, device="cuda") / (Cr * 49) ** 0.5)
            wl = _C.weight_layout(w, 2, Cp, dtype=dtype)
            assert _C.lib().mgn_conv_stem7_blocks(N, H, W, Cp, OH, OW, 64) > 0
            y, part = run(x, wl, OH, OW, stats=True)
            y2 = run(x, wl, OH, OW)
            yo = run(x, wl, OH, OW, old=True)
            ref = torch.nn.functional.conv2d(x[:, :Cr].double(), w.detach().to(dtype).double(), stride=2, padding=3)
            err = float((y.double() - ref).abs().max() / ref.abs().max())
            dold = float((y.float() - yo.float()).abs().max() / yo.float().abs().max())
            st = _C.iabn_from_partials(part, 64, N * OH * OW, None, stats_only=True).double().cpu()
            yd = y.double().permute(1, 0, 2, 3).reshape(64, -1).cpu()
            mean, m2 = yd.mean(1), ((yd - yd.mean(1, keepdim=True)) ** 2).sum(1)
            e1, e2 = float((st[1] - mean).abs().max() / yd.abs().max()), float(((st[2] - m2) / m2).abs().max())
            ok = err < 6e-3 and dold < 8e-3 and torch.equal(y, y2) and e1 < 2e-6 and e2 < 2e-5
            print(f"{str(dtype)[6:]:9s} N{N} {Cr}({Cp})->64 {H}x{W}: vs fp64 {err:.2e} vs packed-tap {dold:.2e} rows {part.shape[0]} "
                  f"mean {e1:.1e} M2 {e2:.1e} {'OK' if ok else 'FAIL'}", flush=True)
            assert ok


def timeit(f, n=10):
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
    B, H, W = 8, 1024, 2048
    for Cr, Cp in [(3, 8), (9, 16)]:
        x = torch.zeros(B, Cp, H, W, device="cuda")
        x[:, :Cr] = torch.randn(B, Cr, H, W, device="cuda")
        x = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = torch.nn.Parameter(torch.randn(64, Cr, 7, 7, device="cuda") / (Cr * 49) ** 0.5)
        wl = _C.weight_layout(w, 2, Cp)
        fl = 2.0 * B * (H // 2) * (W // 2) * 64 * 49 * Cp
        for stats in (False, True):
            to = timeit(lambda: run(x, wl, H // 2, W // 2, old=True, stats=stats))
            tn = timeit(lambda: run(x, wl, H // 2, W // 2, stats=stats))
            print(f"Cp={Cp} stats={stats}: packed-tap {to:7.1f} us ({fl / to / 1e6:5.0f} TF/s padded) | persistent {tn:7.1f} us ({fl / tn / 1e6:5.0f} TF/s, "
                  f"output {B * H * W // 4 * 128 / tn / 1e6:.2f} TB/s)", flush=True)


if __name__ == "__main__":
    check()
    if "--check-only" not in sys.argv:
        bench()
