import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgnet_amd import _C
B, Cin, Cout, H, W, k = 8, 256, 256, 128, 256, 3
x = torch.randn(B, Cin, H, W, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
dy = torch.randn(B, Cout, H, W, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
wo = _C.weight_layout(w, 0)
for _ in range(5):
    _C.conv_igemm(x, wo, (H, W), None, 1, 1)
    _C.conv_wgrad(dy, x, k, k, 1, 1)
torch.cuda.synchronize()
