#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel from p_counter_collection.csv files under a directory."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(root, filt="reproj"):
    acc = defaultdict(lambda: defaultdict(list))
    for f in sorted(glob.glob(os.path.join(root, "*", "*counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if filt not in k:
                continue
            import re
            m = re.search(r"(?:\(anonymous namespace\)::)?(\w+(?:<[^>]*>)?)\(", k.replace("void ", ""))
            short = m.group(1) if m else k[:60]
            acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(k)
        for c, v in sorted(d.items()):
            print(f"   {c:28s} avg {sum(v)/len(v):18.1f}   (n={len(v)})")


if __name__ == "__main__":
    main(sys.argv[1], *(sys.argv[2:]))
