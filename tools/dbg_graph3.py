import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
stage = sys.argv[1]
from bench import synth_batch
from mgnet_amd import _C
B, H, W = 2, 64, (97 if stage == "odd" else 96)
d = synth_batch(B, H, W, 1, torch.device("cuda"))
cfg = _C.make_reproj_cfg(B, H, W, 3)
gl = torch.ones(2, device="cuda")
def step():
    fwd = _C.reproj_loss_fwd(cfg, d["inv"], d["img"], d["prev"], d["nxt"], d["mask"], d["K"], d["poses"], want_grad=True)
    if stage == "fwd_only":
        return fwd["losses"]
    di, dp = _C.reproj_loss_bwd(cfg, d["inv"], d["img"], d["mask"], gl, fwd)
    return dp
step(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = step()
print(stage, "captured"); g.replay(); torch.cuda.synchronize(); print(stage, "replayed OK", out.flatten()[:3])
