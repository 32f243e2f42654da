#!/usr/bin/env python3
"""Does any kernel of the training step read memory it (or a predecessor) never wrote?  Two trainers with identical seeds run the same steps;
for the second one every `torch.empty*` on the GPU is filled with a large finite value first.  Any difference in losses or gradients is an
uninitialised read (the allocator normally hands back recently freed tensors, so such a read is stable in eager steps -- and reads the
PREVIOUS step's data in a launch-plan replay, whose memory is static).  usage: python tools/dbg_uninit.py [fill value] [--locate]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_plan_gpu import _trainer  # noqa: E402

FILL = float(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else 1.0e30
_orig = {n: getattr(torch, n) for n in ("empty", "empty_like", "empty_strided")}
_orig_new_empty = torch.Tensor.new_empty
ACTIVE = [False]
LOG = []


def _poison(t):
    if ACTIVE[0] and isinstance(t, torch.Tensor) and t.is_cuda and t.numel():
        if t.dtype.is_floating_point:
            t.fill_(min(FILL, 6.0e4) if t.dtype == torch.float16 else FILL)
        elif t.dtype == torch.bool:
            t.fill_(True)
        else:
            t.fill_(0x7f)
    return t


def patch():
    for n, f in _orig.items():
        setattr(torch, n, (lambda f: lambda *a, **k: _poison(f(*a, **k)))(f))
    torch.Tensor.new_empty = lambda self, *a, **k: _poison(_orig_new_empty(self, *a, **k))


def run(poison, steps=3):
    ACTIVE[0] = False
    B_, H_, W_ = [int(v) for v in os.environ.get("SIZE", "2x128x256").split("x")]
    tr, batch, _ = _trainer(seed=1, H=H_, W=W_, B=B_)
    out = []
    for k in range(steps):
        ACTIVE[0] = poison
        losses = tr.run_step(batch)
        ACTIVE[0] = False
        torch.cuda.synchronize()
        out.append(({n: float(v) for n, v in losses.items()}, [b["flat_g"].clone() for b in tr.reducer.buckets],
                    {n: p.detach().clone() for n, p in tr.model.named_parameters()}))
    return tr, out


patch()
ta, a = run(False)
tb, b = run(True)
bad = False
for k, ((la, ga, pa), (lb, gb, pb)) in enumerate(zip(a, b)):
    if la != lb:
        bad = True
        print(f"step {k}: losses differ:", {n: (la[n], lb[n]) for n in la if la[n] != lb[n]})
    for i, (x, y) in enumerate(zip(ga, gb)):
        if not torch.equal(x, y):
            bad = True
            d = (x - y).abs()
            print(f"step {k}: gradient bucket {i} differs: {int((d > 0).sum())} of {x.numel()} values, max |diff| {float(d.max()):.3e}, finite {bool(torch.isfinite(y).all())}")
    names = [n for n in pa if not torch.equal(pa[n], pb[n])]
    if names:
        print(f"step {k}: {len(names)} parameters differ after the step, first: {names[:6]}")
    if bad:
        # which parameters' gradients differ (bucket views)
        for bi, bk in enumerate(tb.reducer.buckets):
            for p, o in zip(bk["params"], bk["offsets"]):
                x, y = ga[bi][o:o + p.numel()], gb[bi][o:o + p.numel()]
                if not torch.equal(x, y):
                    nm = [n for n, q in tb.model.named_parameters() if q is p][0]
                    print(f"      grad of {nm} {tuple(p.shape)}: max |diff| {float((x - y).abs().max()):.3e} (max |g| {float(x.abs().max()):.3e})")
        break
print("uninitialised reads found" if bad else f"no difference over {len(a)} steps with torch.empty* filled with {FILL:g}")
