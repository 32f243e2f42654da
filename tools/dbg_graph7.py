import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from bench import synth_batch, HipEvents
from mgnet_amd import _C
B, H, W = 2, 256, 512
d = synth_batch(B, H, W, 1, torch.device("cuda"))
cfg = _C.make_reproj_cfg(B, H, W, 3)
ev = HipEvents(1)
def step():
    return _C.reproj_loss_fwd(cfg, d["inv"], d["img"], d["prev"], d["nxt"], d["mask"], d["K"], d["poses"], want_grad=True)["losses"]
step(); torch.cuda.synchronize()
cfg.prof_begin, cfg.prof_end = ev.pairs[0]
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = step()
print("captured", flush=True)
for _ in range(3):
    g.replay()
torch.cuda.synchronize(); print("replayed OK", out, flush=True)
print("elapsed ms", ev.elapsed_ms(), flush=True)
