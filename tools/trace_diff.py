#!/usr/bin/env python3
"""per-kernel difference of two trace summaries (tools/prof_summary.py output): kernel names with the `_f16` / type suffixes folded, ms per step"""
import re, sys
def load(path, steps):
    d = {}
    for ln in open(path):
        m = re.match(r"^(.{110}) +(\d+) +([\d.]+) +([\d.]+)", ln)
        if not m:
            continue
        name = m.group(1).strip()
        name = re.sub(r"__hip_bfloat16|__half|_Float16", "T16", name)
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        name = re.sub(r"\(.*$", "", name)[:70]
        d[name] = d.get(name, 0.0) + float(m.group(3)) / steps
    return d
a, b, steps = load(sys.argv[1], float(sys.argv[3])), load(sys.argv[2], float(sys.argv[3])), float(sys.argv[3])
rows = sorted(((b.get(k, 0) - a.get(k, 0), k) for k in set(a) | set(b)), reverse=True)
print(f"{'kernel':72s} {'A ms/step':>10s} {'B ms/step':>10s} {'B-A':>8s}")
for dlt, k in rows:
    if abs(dlt) >= 0.01:
        print(f"{k:72s} {a.get(k, 0):10.3f} {b.get(k, 0):10.3f} {dlt:8.3f}")
print(f"{'TOTAL':72s} {sum(a.values()):10.3f} {sum(b.values()):10.3f} {sum(b.values()) - sum(a.values()):8.3f}")
