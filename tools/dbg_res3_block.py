"""The first down-sampling block (res3.0: 64 -> 128 channels, stride 2) evaluated repeatedly on ONE input: which of its stages is not
bit-reproducible?  Stages: conv1 + norm, shortcut + norm, conv2 (raw output + the statistics partials of its epilogue), the
coefficients derived from them, the block output.  CONC=1: the pose encoder's res3.0 runs beside it on a second stream."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mgnet_amd import add_mgnet_config, get_cfg
from mgnet_amd.modeling import ops
from mgnet_amd.registry import build_model

B, H, W = [int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "8x1024x2048").split("x")]
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0")
cfg = get_cfg(); add_mgnet_config(cfg)
cfg.merge_from_file(os.path.join(ROOT, "configs", "bench-c4-cityscapes-videosequence.yaml"))
cfg.merge_from_list(["MODEL.DEVICE", str(dev), "SOLVER.IMS_PER_BATCH", B])
torch.manual_seed(0)
model = build_model(cfg).train()
blk, blk2 = model.backbone.res3[0], model.pose_net.pose_encoder.res3[0]
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.randn(B, 64, H // 4, W // 4, device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_()
x2 = (x.detach() * 0.5 + 0.1).requires_grad_()
side = torch.cuda.Stream()


def once(b, xin):
    out, xsub = b.conv1(xin, with_skip=2)
    sc = b.shortcut(xsub, full=xin)
    c2 = b.conv2
    raw = ops.conv2d(out, c2.weight, c2.bias, c2.stride, c2.padding, stats_for=c2.norm)
    st = raw.__dict__.get("_mgn_stats")
    part = None if st is None else st[0].clone()
    rawc = raw.detach().clone()
    y = ops.abn_add_relu(raw, c2.norm, sc)
    return {"conv1+norm": out.detach().clone(), "shortcut+norm": sc.detach().clone(), "conv2 raw": rawc, "conv2 partials": part, "block out": y.detach().clone(),
            "running_mean conv2": c2.norm.running_mean.clone()}


def run():
    if os.environ.get("CONC") == "1":
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            once(blk2, x2)
    r = once(blk, x)
    torch.cuda.synchronize()
    return r


ref = run()
print({k: (None if v is None else tuple(v.shape)) for k, v in ref.items()})
for rep in range(REPS):
    cur = run()
    msg = []
    for k in ref:
        if ref[k] is None or k.startswith("running"):
            continue
        a, b = ref[k].float(), cur[k].float()
        if not torch.equal(ref[k], cur[k]):
            nz = (a != b)
            idx = nz.nonzero()
            msg.append(f"{k}: {int(nz.sum())} elements, max |d| {float((a - b).abs().max()):.3e}, first at {idx[0].tolist()}, last at {idx[-1].tolist()}")
    print(f"[rep {rep}] " + ("identical" if not msg else " | ".join(msg)), flush=True)
