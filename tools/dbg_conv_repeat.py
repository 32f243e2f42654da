"""Inside the real training-mode forward pass at the benchmark's size: every ops.conv2d call is evaluated THREE times on the spot from the
same input / weights (no synchronisation: the other trunk's stream keeps running beside it) and the results are compared after the pass.
A convolution whose repeated evaluations differ is nondeterministic under concurrency; QUIET=1 synchronises the device before each
repeated call (no concurrency).  Usage: dbg_conv_repeat.py [BxHxW]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mgnet_amd import add_mgnet_config, get_cfg
from mgnet_amd.data import synthetic_batch
from mgnet_amd.engine import Trainer
from mgnet_amd.modeling import ops
from mgnet_amd.registry import build_model

B, H, W = [int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "8x1024x2048").split("x")]
dev = torch.device("cuda:0")
cfg = get_cfg(); add_mgnet_config(cfg)
cfg.merge_from_file(os.path.join(ROOT, "configs", "bench-c4-cityscapes-videosequence.yaml"))
cfg.merge_from_list(["MODEL.DEVICE", str(dev), "SOLVER.IMS_PER_BATCH", B])
torch.manual_seed(0)
model = build_model(cfg)
tr = Trainer(cfg, model)
batch = synthetic_batch(B, H, W, dev, seed=1234)
for _ in range(2):
    tr.run_step(batch)
torch.cuda.synchronize()
model.train()
real = ops.conv2d
found = []
QUIET = os.environ.get("QUIET") == "1"


def conv2d(x, weight, bias=None, stride=1, padding=0, relu=False, with_skip=False, stats_for=None, keep_pad=False):
    outs = []
    with torch.no_grad():
        for _ in range(2):
            if QUIET:
                torch.cuda.synchronize()
            y = real(x.detach(), weight.detach(), None if bias is None else bias.detach(), stride, padding, relu=relu)
            outs.append(y.t if isinstance(y, ops.PaddedMap) else y)
    if QUIET:
        torch.cuda.synchronize()
    r = real(x, weight, bias, stride, padding, relu=relu, with_skip=with_skip, stats_for=stats_for, keep_pad=keep_pad)
    y = r[0] if isinstance(r, tuple) else r
    y = y.t if isinstance(y, ops.PaddedMap) else y
    found.append((f"x{tuple(x.shape)} w{tuple(weight.shape)} s{stride} p{padding} stream {torch.cuda.current_stream().cuda_stream:#x}",
                  outs[0], outs[1], y.detach().clone()))
    return r


ops.conv2d = conv2d
import mgnet_amd.modeling.layers as L, mgnet_amd.modeling.res_net as R
assert L.ops is ops and R.ops is ops
for rep in range(2):
    found.clear()
    with torch.no_grad():
        pass
    tr.reducer.zero_grad()
    with tr.storage:
        ld = model(batch)
    torch.cuda.synchronize()
    bad = 0
    for k, (name, a, b, c) in enumerate(found):
        c = c[:, :a.shape[1]] if c.shape != a.shape and c.dim() == 4 else c
        same_ab, same_ac = torch.equal(a, b), (a.shape == c.shape and torch.equal(a, c))
        if not (same_ab and same_ac):
            bad += 1
            d1 = int((a != b).sum())
            d2 = int((a != c).sum()) if a.shape == c.shape else -1
            print(f"[rep {rep}] conv #{k} {name}: repeated evaluations differ: {d1} elements (1 vs 2), {d2} (1 vs the call itself), of {a.numel()}", flush=True)
    print(f"[rep {rep}] {bad} of {len(found)} convolutions not reproducible ({'no concurrency' if QUIET else 'with the other streams running'})", flush=True)
    del ld
