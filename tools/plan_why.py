#!/usr/bin/env python3
"""Why does a launch of the recorded step wait for another stream?  For the largest cross-stream waits of a traced replay: the memory
ranges the waiting launch shares with the launch that released it (engine/plan.py derives the cross-stream order from exactly these).
usage: MGN_PLAN_DEBUG=1 python tools/plan_why.py [kernel-name substring of the waiting launch]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ["MGN_PLAN_DEBUG"] = "1"
import critical_path as cp  # noqa: E402


def main():
    import argparse
    args = argparse.Namespace(batch=8, height=1024, width=2048, dtype="bf16")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    trainer, batch = cp.build_trainer(args, dev)
    for _ in range(5):
        trainer.run_step(batch)
    plan = trainer.record_plan(batch)
    for _ in range(3):
        trainer.replay_plan()
    plan.trace(True)
    for _ in range(3):
        trainer.replay_plan()
    torch.cuda.synchronize()
    b, e = plan.trace_read()
    items, dbg = plan.items, plan.debug_items
    cp.demangle([it["name"] for it in items])
    preds = cp.schedule_edges(plan)
    node = [it["node"] for it in items]
    ok = [it["kind"] == 0 and not np.isnan(e[it["node"]]) for it in items]
    main_st = plan.main.cuda_stream
    sname = lambda st: "main" if st == main_st else hex(st)[-4:]
    rows = []
    for k, it in enumerate(items):
        if not ok[k]:
            continue
        same = [p for p in preds[k] if items[p]["stream"] == it["stream"] and ok[p]]
        cross = [p for p in preds[k] if items[p]["stream"] != it["stream"] and ok[p]]
        if not cross:
            continue
        r = max(cross, key=lambda p: e[node[p]])
        ready_same = max([e[node[p]] for p in same], default=0.0)
        wait = e[node[r]] - ready_same
        if wait > 0.05:
            rows.append((wait, k, r))
    rows.sort(reverse=True)
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    for wait, k, r in rows[:12]:
        if pat and pat not in cp.short(items[k]["name"]):
            continue
        mine = [(a, c, 0) for a, c in dbg[k]["reads"]] + [(a, c, 1) for a, c in dbg[k]["writes"]]
        theirs = [(a, c, 0) for a, c in dbg[r]["reads"]] + [(a, c, 1) for a, c in dbg[r]["writes"]]
        shared = sorted({(max(a, a2), min(c, c2), w, w2) for a, c, w in mine for a2, c2, w2 in theirs if a < c2 and a2 < c and (w or w2)})
        print(f"{wait * 1e3:8.0f} us  {cp.short(items[k]['name'])}@{sname(items[k]['stream'])} (item {k}) waits for {cp.short(items[r]['name'])}@{sname(items[r]['stream'])} (item {r})")
        if not shared:
            print("            no directly shared range: the edge is implied by an EARLIER launch of that stream (the event of the latest needed position covers it)")
            # find the earlier item on r's stream that k really conflicts with
            for q in range(r, -1, -1):
                if items[q]["stream"] != items[r]["stream"] or items[q]["kind"] != 0 and not dbg[q]["writes"] and not dbg[q]["reads"]:
                    continue
                th = [(a, c, 0) for a, c in dbg[q]["reads"]] + [(a, c, 1) for a, c in dbg[q]["writes"]]
                sh = sorted({(max(a, a2), min(c, c2), w, w2) for a, c, w in mine for a2, c2, w2 in th if a < c2 and a2 < c and (w or w2)})
                if sh:
                    print(f"            -> real conflict with item {q} {cp.short(items[q]['name'])} (ended {e[node[q]] if items[q]['kind']==0 else float('nan'):.3f} ms; the releasing launch ended {e[node[r]]:.3f})")
                    shared = sh
                    break
        for a, c, w, w2 in shared[:4]:
            print(f"            range {a:#x}..{c:#x} ({(c - a) / 1024:.1f} KiB): this launch {'writes' if w else 'reads'}, the other {'writes' if w2 else 'reads'}")
            for tag, j in (("this ", k), ("other", r)):
                for ai, kind, pp, blk in plan.arg_debug.get(items[j]["node"], []):
                    if blk is not None and blk[0] < c and a < blk[1]:
                        print(f"                {tag}: argument {ai} ({kind}) = {pp:#x} = block start + {(pp - blk[0]) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
