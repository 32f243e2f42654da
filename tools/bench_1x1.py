"""A/B of the implicit-GEMM tile choice on the memory-bound 1x1 layers (run with MGN_CONV_BIG=0 / unset)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgnet_amd import _C
B = 8
def cl(*s):
    return torch.randn(*s, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
cases = [("1x1 256->256 @128x256", cl(B, 256, 128, 256), torch.randn(256, 256, 1, 1, device="cuda") * 0.05, (128, 256), 0),
         ("1x1 512->256 @32x64", cl(B, 512, 32, 64), torch.randn(256, 512, 1, 1, device="cuda") * 0.05, (32, 64), 0),
         ("3x3 256->256 @64x128", cl(B, 256, 64, 128), torch.randn(256, 256, 3, 3, device="cuda") * 0.05, (64, 128), 1),
         ("3x3 256->256 @32x64", cl(B, 256, 32, 64), torch.randn(256, 256, 3, 3, device="cuda") * 0.05, (32, 64), 1),
         ("3x3 512->512 @32x64", cl(B, 512, 32, 64), torch.randn(512, 512, 3, 3, device="cuda") * 0.05, (32, 64), 1)]
for name, x, w, osz, pad in cases:
    wl = _C.weight_layout(torch.nn.Parameter(w), 0)
    for _ in range(5):
        _C.conv_igemm(x, wl, osz, None, 1, pad)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        _C.conv_igemm(x, wl, osz, None, 1, pad)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    fl = 2.0 * x.shape[0] * osz[0] * osz[1] * w.shape[0] * w.shape[1] * w.shape[2] * w.shape[3]
    print(f"{name:28s} {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s   MGN_CONV_BIG={os.environ.get('MGN_CONV_BIG')}")
