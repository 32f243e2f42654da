"""Which module's output is the first that differs between two evaluations of the SAME training-mode forward pass (same parameters, same
batch)?  Forward hooks clone every leaf-ish module's output on the stream it was produced on; the two passes are compared in call order.
Then the same for the backward pass: parameter gradients in reverse registration order.  Usage: dbg_fwd_bisect.py [BxHxW] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mgnet_amd import add_mgnet_config, get_cfg
from mgnet_amd.data import synthetic_batch
from mgnet_amd.engine import Trainer
from mgnet_amd.registry import build_model

B, H, W = [int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "8x1024x2048").split("x")]
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
cfg = get_cfg(); add_mgnet_config(cfg)
cfg.merge_from_file(os.path.join(ROOT, "configs", "bench-c4-cityscapes-videosequence.yaml"))
cfg.merge_from_list(["MODEL.DEVICE", str(dev), "SOLVER.IMS_PER_BATCH", B, "MODEL.SEM_SEG_HEAD.OHEM_N_MIN", min(524287, B * H * W // 4 - 1)])
torch.manual_seed(0)
model = build_model(cfg)
tr = Trainer(cfg, model)
batch = synthetic_batch(B, H, W, dev, seed=1234)
for _ in range(2):
    tr.run_step(batch)
torch.cuda.synchronize()
model.train()
log = []


def tensors(o):
    if isinstance(o, torch.Tensor):
        return [o]
    if hasattr(o, "t") and isinstance(getattr(o, "t"), torch.Tensor):   # ops.PaddedMap
        return [o.t]
    if isinstance(o, (list, tuple)):
        return [t for x in o for t in tensors(x)]
    if isinstance(o, dict):
        return [t for x in o.values() for t in tensors(x)]
    return []


def hook(name):
    def f(mod, inp, out):
        log.append((name, [t.detach().clone() for t in tensors(out) if t.is_cuda]))
    return f


for n, m in model.named_modules():
    if n:
        m.register_forward_hook(hook(n))

# the fused block tail and the statistics finalisation are not module calls: log their operands and results too
from mgnet_amd import _C
from mgnet_amd.modeling import ops
_tail, _fp, _cnt = ops.abn_add_relu, _C.iabn_from_partials, [0]


def tail(x, norm, shortcut):
    k = _cnt[0]; _cnt[0] += 1
    st = x.__dict__.get("_mgn_stats")
    log.append((f"tail#{k} conv2 raw", [x.detach().clone()]))
    if st is not None:
        log.append((f"tail#{k} partials", [st[0].detach().clone()] + ([] if st[1] is None else [st[1].detach().clone()])))
    log.append((f"tail#{k} shortcut", [shortcut.detach().clone()]))
    y = _tail(x, norm, shortcut)
    log.append((f"tail#{k} out", [y.detach().clone()]))
    return y


def from_partials(*a, **k):
    out = _fp(*a, **k)
    log.append((f"from_partials rows={a[0].shape[0]} C={a[1]}", [out.detach().clone()]))
    return out


ops.abn_add_relu = tail
_C.iabn_from_partials = from_partials
import mgnet_amd.modeling.res_net as _rn
assert _rn.ops is ops
names = [n for n, p in model.named_parameters() if p.requires_grad]
params = dict(model.named_parameters())


buf0 = [b.detach().clone() for b in model.buffers()]


def once():
    log.clear()
    _cnt[0] = 0
    with torch.no_grad():   # (the running statistics are the shift of the fused statistics epilogues: same state for every evaluation)
        for b, q in zip(model.buffers(), buf0):
            b.copy_(q)
    tr.reducer.zero_grad()
    with tr.storage:
        ld = model(batch)
        tr._backward(ld)
    tr.reducer.finish()
    torch.cuda.synchronize()
    return list(log), {k: v.detach().clone() for k, v in ld.items()}, {n: params[n].grad.detach().clone() for n in names}


ref = once()
for rep in range(REPS):
    cur = once()
    assert [a[0] for a in ref[0]] == [a[0] for a in cur[0]]
    bad = []
    for (n, ta), (_, tb) in zip(ref[0], cur[0]):
        d = [float((x.float() - y.float()).abs().max()) for x, y in zip(ta, tb) if not torch.equal(x, y)]
        if d:
            nd = [int((x != y).sum()) for x, y in zip(ta, tb)]
            bad.append((n, max(d), nd))
    print(f"[rep {rep}] forward: {len(bad)} of {len(cur[0])} module outputs differ" + ("" if not bad else "; first: " + "; ".join(f"{n} (max |d| {d:.3e}, elements {nd})" for n, d, nd in bad[:6])), flush=True)
    dl = {k: (float(ref[1][k]), float(cur[1][k])) for k in ref[1] if not torch.equal(ref[1][k], cur[1][k])}
    gb = [(n, float((cur[2][n] - ref[2][n]).abs().max()), float(ref[2][n].abs().max())) for n in names if not torch.equal(cur[2][n], ref[2][n])]
    print(f"[rep {rep}] losses differing: {dl}; {len(gb)} of {len(names)} gradients differ" + ("" if not gb else "; LAST in registration order (first computed): " +
          "; ".join(f"{n} {d:.2e}/{m:.2e}" for n, d, m in gb[-5:])), flush=True)
