import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
stage = sys.argv[1]
from bench import synth_batch
from mgnet_amd import _C
B, H, W = 2, 64, 96
d = synth_batch(B, H, W, 1, torch.device("cuda"))
inv = [x.requires_grad_(True) for x in d["inv"]]
poses = d["poses"].requires_grad_(True)
cfg = _C.make_reproj_cfg(B, H, W, 3)
class Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cfg, img, prev, nxt, mask, cam, poses, *inv):
        inv = [t.contiguous() for t in inv]
        if stage == "nofwdkernel":
            losses = torch.ones(2, device=img.device)
            ctx.g = [torch.ones_like(t) for t in inv]; ctx.dp = torch.ones_like(poses)
            return losses
        fwd = _C.reproj_loss_fwd(cfg, inv, img, prev, nxt, mask, cam, poses.contiguous(), want_grad=(stage != "nograd"))
        if stage == "nograd":
            ctx.g = [torch.ones_like(t) for t in inv]; ctx.dp = torch.ones_like(poses)
        else:
            ctx.g, ctx.dp = fwd["g_inv"], fwd["d_pose"]
        if stage == "holdws":
            ctx.ws = fwd["workspace"]
        if stage == "clone_out":
            return fwd["losses"].clone()
        return fwd["losses"]
    @staticmethod
    def backward(ctx, gl):
        if stage == "freshgrads":
            return (None,) * 6 + (torch.zeros_like(ctx.dp),) + tuple(torch.zeros_like(t) for t in ctx.g)
        return (None,) * 6 + (ctx.dp,) + tuple(ctx.g)
def step():
    losses = Fn.apply(cfg, d["img"], d["prev"], d["nxt"], d["mask"], d["K"], poses, *inv)
    losses.sum().backward()
    return losses
step(); torch.cuda.synchronize()
for x in inv: x.grad = None
poses.grad = None
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = step()
print(stage, "captured"); g.replay(); torch.cuda.synchronize(); print(stage, "replayed OK", out)
