"""a few launches of the stride-2 data-gradient window kernel on the three C4 shapes (for rocprofv3 --pmc)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgnet_amd import _C
for Cin, Cout, H, W in [(64, 128, 256, 512), (128, 256, 128, 256), (256, 512, 64, 128)]:
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") / (Cin * 9) ** 0.5
    dy = torch.randn(8, Cout, H // 2, W // 2, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wl = _C._weight_layout_now(w, 1, 0, None, 0, torch.bfloat16)
    for _ in range(4):
        _C.conv_igemm(dy, wl, (H, W), None, 1, 1, up=2)
torch.cuda.synchronize()
