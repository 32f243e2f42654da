"""a few launches of the semantic-head loss kernels at the benchmark shape (for rocprofv3 --pmc / --kernel-trace)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgnet_amd import _C
B, K, h, w, H, W = 8, 19, 128, 256, 1024, 2048
torch.manual_seed(0)
lr = torch.randn(B, 24, h, w, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)[:, :K]
labels = torch.randint(0, K, (B, H, W), device="cuda")
labels[torch.rand(B, H, W, device="cuda") < 0.05] = 255
weights = torch.rand(B, H, W, device="cuda") + 0.5
g = torch.ones(1, device="cuda")
for _ in range(4):
    ce, sums = _C.upce_fwd(lr, labels, weights, H, W, 255, 0.3567)
    sel, loss = _C.ohem_select(ce, sums, 0.3567, B * H * W // 16, False)
    dlg = _C.upce_bwd(lr, labels, weights, H, W, 255, ce, sel.float().contiguous(), g, 24)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    dlg = _C.upce_bwd(lr, labels, weights, H, W, 255, ce, sel.float().contiguous(), g, 24)
e1.record()
torch.cuda.synchronize()
print(f"upce_bwd {e0.elapsed_time(e1) / 10 * 1e3:.1f} us  loss {float(loss):.4f}")
