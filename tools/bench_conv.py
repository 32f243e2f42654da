#!/usr/bin/env python3
"""Per-layer timing of the HIP conv kernels vs MIOpen (torch) on MGNet layer shapes (B=8, 1024x2048 input)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from mgnet_amd import _C

SHAPES = [  # name, Cin, Cout, H, W (input), k, s
    ("head3x3 256->256 /8", 256, 256, 128, 256, 3, 1),
    ("res2 3x3 64->64 /4", 64, 64, 256, 512, 3, 1),
    ("res3 3x3 128->128 /8", 128, 128, 128, 256, 3, 1),
    ("res4 3x3 256->256 /16", 256, 256, 64, 128, 3, 1),
    ("res5 3x3 512->512 /32", 512, 512, 32, 64, 3, 1),
    ("res3 3x3s2 64->128", 64, 128, 256, 512, 3, 2),
    ("ffm 1x1 256->256 /8", 256, 256, 128, 256, 1, 1),
]
B = int(os.environ.get("B", 8))


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


for name, Cin, Cout, H, W, k, s in SHAPES:
    p = k // 2
    x = torch.randn(B, Cin, H, W, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    dy = torch.randn(B, Cout, OH, OW, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wo = w.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)
    wt = w.flip(2, 3).permute(1, 2, 3, 0).contiguous().to(torch.bfloat16)
    gf = 2.0 * B * OH * OW * Cout * Cin * k * k / 1e9
    t_f = timeit(lambda: _C.conv_igemm(x, wo, (OH, OW), None, s, p))
    t_d = timeit(lambda: _C.conv_igemm(dy, wt, (H, W), None, 1, k - 1 - p, up=s))
    t_w = timeit(lambda: _C.conv_wgrad(dy, x, k, k, s, p))  # incl. zero-init of dw
    wb = w.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    xr = x.detach().requires_grad_(True); wr = wb.detach().requires_grad_(True)
    t_tf = timeit(lambda: F.conv2d(x, wb, None, s, p))
    y = F.conv2d(xr, wr, None, s, p)
    t_tb = timeit(lambda: torch.autograd.grad(y, [xr, wr], dy, retain_graph=True))
    print(f"{name:24s} {gf:7.1f} GF | hip fwd {t_f:6.3f} ms {gf/t_f:6.0f} TF  dgrad {t_d:6.3f} {gf/t_d:6.0f}  wgrad {t_w:6.3f} {gf/t_w:6.0f} | "
          f"miopen fwd {t_tf:6.3f} {gf/t_tf:6.0f}  bwd(d+w) {t_tb:6.3f} {2*gf/t_tb:6.0f}", flush=True)
