"""rows_per_wave sweep of the fused reprojection kernel (loss + gradient) at the C4 shape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mgnet_amd import _C
from mgnet_amd.modeling.loss import _ReprojLossFn
B, H, W = 8, 1024, 2048
d = bench.synth_batch(B, H, W, 1234, torch.device("cuda"))
inv = [x.requires_grad_(True) for x in d["inv"]]
poses = d["poses"].requires_grad_(True)
w = torch.ones(2, device="cuda")
for rh in (0, 16, 32, 64, 128, 256):
    cfg = _C.make_reproj_cfg(B, H, W, 3, rows_per_wave=rh)
    def step():
        losses = _ReprojLossFn.apply(cfg, d["img"], d["prev"], d["nxt"], d["mask"], d["K"], poses, *inv)
        (losses * w).sum().backward()
        for x in inv:
            x.grad = None
        poses.grad = None
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        step()
    e1.record()
    torch.cuda.synchronize()
    print(f"rows_per_wave {rh:4d}: {e0.elapsed_time(e1) / 10:.3f} ms per fwd+bwd")
