import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
stage = sys.argv[1]
from bench import synth_batch
from mgnet_amd import _C
from mgnet_amd.modeling.loss import _ReprojLossFn
B, H, W = 2, 64, 96
d = synth_batch(B, H, W, 1, torch.device("cuda"))
inv = [x.requires_grad_(True) for x in d["inv"]]
poses = d["poses"].requires_grad_(True)
cfg = _C.make_reproj_cfg(B, H, W, 3)
def step():
    losses = _ReprojLossFn.apply(cfg, d["img"], d["prev"], d["nxt"], d["mask"], d["K"], poses, *inv)
    if stage != "reproj_fwd":
        losses.sum().backward()
    return losses
step(); torch.cuda.synchronize()
for x in inv: x.grad = None
poses.grad = None
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = step()
print(stage, "captured"); g.replay(); torch.cuda.synchronize(); print(stage, "replayed OK", out)
