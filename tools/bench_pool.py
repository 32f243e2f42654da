import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgnet_amd import _C
x = torch.randn(8, 64, 512, 1024, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
sc = torch.rand(64, device="cuda") + 0.5; of = torch.randn(64, device="cuda") * 0.1
for _ in range(3): y, arg = _C.abn_maxpool_fwd(x, sc, of, 1, 0.01)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): y, arg = _C.abn_maxpool_fwd(x, sc, of, 1, 0.01)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10 * 1e3
print(f"abn_maxpool_fwd {t:.1f} us  {(x.numel()*2 + y.numel()*3) / t / 1e6:.2f} TB/s")
# backward: dx from (x, d pooled, argmax) with the norm's backward folded in
w, b = torch.rand(64, device="cuda") + 0.5, torch.randn(64, device="cuda") * 0.1
coef = torch.stack([sc, of, torch.zeros(64, device="cuda"), torch.ones(64, device="cuda")])
dpool = torch.randn_like(y)
sums = torch.randn(2, 64, device="cuda")
f = lambda: _C.abn_maxpool_bwd(x, dpool, arg, coef, w, b, sums, float(x.numel() // 64), 1e-5, 1, 0.01)
for _ in range(3): dx = f()
e0.record()
for _ in range(10): dx = f()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10 * 1e3
print(f"abn_maxpool_bwd {t:.1f} us  {(x.numel()*4 + y.numel()*3) / t / 1e6:.2f} TB/s")
