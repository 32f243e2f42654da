"""a few launches of the windowed 3x3 kernel on the benchmark's shapes (for rocprofv3 --pmc / --kernel-trace)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgnet_amd import _C
B = 8
for (Cin, Cout, H, W, pr) in [(256, 256, 128, 256, 16), (128, 128, 128, 256, 16), (512, 512, 32, 64, 8)]:
    x = torch.randn(B, Cin, H, W, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wl = (torch.randn(Cout, 3, 3, Cin, device="cuda") / (Cin * 9) ** 0.5).to(torch.bfloat16)
    for _ in range(4):
        _C.conv3x3_win(x, wl, patch_rows=pr)
torch.cuda.synchronize()
