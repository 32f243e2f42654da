#!/usr/bin/env python3
"""Per-kernel table from `rocprofv3 --pmc` passes over a few bench steps: LDS bank-conflict share, VALU-active share of the wave
cycles, waiting share -- sorted by wave cycles.  usage: pmc_step_table.py <dir with pass subdirs>"""
import csv, glob, os, re, sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "")
        m = re.search(r"(?:\(anonymous namespace\)::)?([\w:]+(?:<[^>]*>)?)\(", k)
        short = (m.group(1) if m else k)[:44]
        acc[short][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[short][r["Counter_Name"]] += 1
rows = []
for k, d in acc.items():
    wc = d.get("SQ_WAVE_CYCLES", 0.0)
    if wc <= 0:
        continue
    n = max(cnt[k].values())
    rows.append((wc, k, n, d))
rows.sort(reverse=True)
print(f"{'kernel':44s} {'calls':>6s} {'wave_cyc(M)':>11s} {'valu%':>6s} {'wait%':>6s} {'lds_act%':>8s} {'bankconf/lds_act':>16s} {'valu/wave':>9s} {'lds/wave':>8s}")
for wc, k, n, d in rows[:45]:
    g = lambda c: d.get(c, 0.0)
    waves = g("SQ_WAVES") or 1.0
    print(f"{k:44s} {n:6d} {wc / 1e6:11.1f} {100 * g('SQ_ACTIVE_INST_VALU') / wc:6.1f} {100 * g('SQ_WAIT_ANY') / wc:6.1f} "
          f"{100 * g('SQ_LDS_IDX_ACTIVE') / wc:8.1f} {(g('SQ_LDS_BANK_CONFLICT') / g('SQ_LDS_IDX_ACTIVE')) if g('SQ_LDS_IDX_ACTIVE') else 0:16.2f} "
          f"{g('SQ_INSTS_VALU') / waves:9.0f} {g('SQ_INSTS_LDS') / waves:8.0f}")
