"""Which kernel arguments of the recorded step point OUTSIDE every allocator block the recorder knows (engine/plan.py treats such a pointer as
library-owned memory and orders launches only on the exact address)?  Lists them per kernel name and argument."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ["MGN_PLAN_DEBUG"] = "1"
import torch
import critical_path as cp
import argparse
B, H, W = [int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "8x1024x2048").split("x")]
args = argparse.Namespace(batch=B, height=H, width=W, dtype="bf16")
dev = torch.device("cuda", 0)
trainer, batch = cp.build_trainer(args, dev)
for _ in range(4):
    trainer.run_step(batch)
plan = trainer.record_plan(batch)
cp.demangle([it["name"] for it in plan.items])
name_of = {it["node"]: cp.short(it["name"]) for it in plan.items if it["kind"] == 0}
seen = {}
for node, ents in plan.arg_debug.items():
    for (k, kind, p, blk) in ents:
        if blk is None and p:
            seen.setdefault((name_of.get(node, "?"), k, kind), set()).add(p)
for (nm, k, kind), ps in sorted(seen.items()):
    print(f"{nm:40s} arg {k:2d} {kind:8s}: {len(ps)} distinct unresolved pointers, e.g. {min(ps):#x}")
print("unresolved in total:", len(plan._unresolved), "| report:", plan.report)
