"""The torch ops a recorded step still contains (host-issued closures of the launch plan): name, stream, tensor shapes / dtypes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import argparse
import torch
import critical_path as cp
B, H, W = [int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "8x1024x2048").split("x")]
tr, batch = cp.build_trainer(argparse.Namespace(batch=B, height=H, width=W, dtype="bf16"), torch.device("cuda", 0))
for _ in range(4):
    tr.run_step(batch)
plan = tr.record_plan(batch)
sid = {st: i for i, st in enumerate(sorted({it["stream"] for it in plan.items}))}
k = 0
for i, it in enumerate(plan.items):
    if it["kind"] == 1:
        c = plan.closures[k]; k += 1
        f, a, kw = c["call"]
        def sh(x):
            if isinstance(x, torch.Tensor):
                return f"{tuple(x.shape)}:{str(x.dtype).replace('torch.', '')}{'' if x.is_contiguous() else ':nc'}"
            if isinstance(x, (list, tuple)):
                return "[" + ", ".join(sh(y) for y in x[:4]) + (", ..." if len(x) > 4 else "") + "]"
            return repr(x)[:20]
        mb = sum(t.numel() * t.element_size() for t in list(a) + list(kw.values()) if isinstance(t, torch.Tensor)) / 1e6
        print(f"#{i:3d} s{sid[it['stream']]} {c['name']:24s} {mb:8.1f} MB  args " + ", ".join(sh(x) for x in a)[:150] + ("  out " + sh(kw.get("out")) if "out" in kw else ""))
