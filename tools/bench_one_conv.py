import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgnet_amd import _C
B = 8
def cl(*s):
    return torch.randn(*s, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
x256, dy256 = cl(B, 256, 128, 256), cl(B, 256, 128, 256)
w256 = torch.nn.Parameter(torch.randn(256, 256, 3, 3, device="cuda") * 0.05)
x64, dy64 = cl(B, 64, 256, 512), cl(B, 64, 256, 512)
w64 = torch.nn.Parameter(torch.randn(64, 64, 3, 3, device="cuda") * 0.05)
xs, dys = cl(B, 16, 1024, 2048), cl(B, 64, 512, 1024)
for _ in range(5):
    _C.conv_igemm(x256, _C.weight_layout(w256, 0), (128, 256), None, 1, 1)      # conv_igemm_big256
    _C.conv_wgrad(dy256, x256, 3, 3, 1, 1)                                        # conv_wgrad3x3_s128 (16 tiles)
    _C.conv_igemm(x64, _C.weight_layout(w64, 0), (256, 512), None, 1, 1)         # conv3x3_c64
    _C.conv_wgrad(dy64, x64, 3, 3, 1, 1)                                          # conv_wgrad3x3_s128 (1 tile)
    _C.conv_wgrad(dys, xs, 7, 7, 2, 3, cin_real=9)                                # conv_wgrad_stem16
torch.cuda.synchronize()
