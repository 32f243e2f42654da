"""a few launches of the persistent stem kernels on the benchmark's shapes (for rocprofv3 --pmc / --kernel-trace)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgnet_amd import _C
B, H, W = 8, 1024, 2048
for Cr, Cp in [(3, 8), (9, 16)]:
    x = torch.zeros(B, Cp, H, W, device="cuda")
    x[:, :Cr] = torch.randn(B, Cr, H, W, device="cuda")
    x = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(64, Cr, 7, 7, device="cuda") / (Cr * 49) ** 0.5)
    wl = _C.weight_layout(w, 2, Cp)
    for _ in range(4):
        holder = []
        _C.conv_igemm(x, wl, (H // 2, W // 2), None, 2, 3, khw=(7, 7), stats=(None, holder))
torch.cuda.synchronize()
