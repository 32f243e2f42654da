#!/usr/bin/env python3
"""Independent check of a recorded step's replay schedule (engine/plan.py derive_schedule): for EVERY pair of items on different streams
whose memory ranges overlap with at least one writer, the later one must be ordered behind the earlier one by the schedule's own
happens-before relation (stream order + event record -> wait).  Reports unordered pairs.  usage: plan_check_edges.py [BxHxW]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ["MGN_PLAN_DEBUG"] = "1"
import argparse
import numpy as np
import torch
import critical_path as cp
from mgnet_amd.engine import plan as P

B, H, W = [int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "4x512x1024").split("x")]
args = argparse.Namespace(batch=B, height=H, width=W, dtype="bf16")
dev = torch.device("cuda", 0)
trainer, batch = cp.build_trainer(args, dev)
for _ in range(4):
    trainer.run_step(batch)
plan = trainer.record_plan(batch)
items = plan.debug_items
n = len(items)
cp.demangle([it["name"] for it in items])
# happens-before from the op list: walk the ops, per stream the last item seen; RECORD binds an event to (stream's last item);
# WAIT adds a pending predecessor to the stream's next item
ops = plan.ops
streams = sorted({it["stream"] for it in items})
sid = {s: k for k, s in enumerate(streams)}
last = {s: -1 for s in streams}
pending = {s: [] for s in streams}
ev_item = {}
preds = [[] for _ in range(n)]
k = -1
order = []   # item index per LAUNCH/BREAK op in op order
it_iter = iter(range(n))
for (typ, a, st) in ops:
    if typ in (P.LAUNCH, P.BREAK):
        k = next(it_iter)
        s = items[k]["stream"]
        if last[s] >= 0:
            preds[k].append(last[s])
        preds[k].extend(pending[s]); pending[s] = []
        last[s] = k
    elif typ == P.RECORD:
        ev_item[a] = last.get(st, -1)
    elif typ == P.WAIT:
        if a in ev_item and ev_item[a] >= 0:
            pending[st].append(ev_item[a])
# reachability (bitsets as python ints)
reach = [0] * n
for i in range(n):
    r = 0
    for p in preds[i]:
        r |= reach[p] | (1 << p)
    reach[i] = r
bad = []
acc = []
for i, it in enumerate(items):
    acc.append([(a, b, 0) for a, b in it["reads"]] + [(a, b, 1) for a, b in it["writes"]])
for j in range(n):
    for i in range(j):
        if items[i]["stream"] == items[j]["stream"] or (reach[j] >> i) & 1:
            continue
        hit = None
        for (a, b, w) in acc[j]:
            for (c, d, v) in acc[i]:
                if a < d and c < b and (w or v) and not (b - a == 1 and a == 1):
                    hit = (a, b, w, c, d, v); break
            if hit: break
        if hit:
            bad.append((i, j, hit))
print(f"{n} items, {sum(1 for o in ops if o[0] in (P.RECORD,))} event records; unordered conflicting pairs: {len(bad)}")
for i, j, h in bad[:40]:
    print(f"  {i} [{sid[items[i]['stream']]}] {cp.short(items[i]['name'])[:40]:40s} -> {j} [{sid[items[j]['stream']]}] {cp.short(items[j]['name'])[:40]:40s} range {h[0]:#x}+{h[1]-h[0]} ({'W' if h[2] else 'R'}) vs {h[3]:#x}+{h[4]-h[3]} ({'W' if h[5] else 'R'})")
