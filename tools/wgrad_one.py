"""a few launches of the 3x3 weight-gradient kernel on the benchmark's largest shape (for rocprofv3 --pmc); ROOT=<tree> picks the tree"""
import os, sys, torch
sys.path.insert(0, os.environ.get("MGN_TREE") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgnet_amd import _C
B = 8
for (Cin, Cout, H, W) in [(256, 256, 128, 256), (128, 128, 128, 256)]:
    x = torch.randn(B, Cin, H, W, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, Cout, H, W, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    for _ in range(4):
        _C.conv_wgrad(dy, x, 3, 3, 1, 1)
torch.cuda.synchronize()
