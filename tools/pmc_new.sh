#!/bin/bash
# usage: tools/pmc_new.sh <tag>  -- FETCH_SIZE / WRITE_SIZE passes (separate runs, no tracing) for the f1/f2 kernels
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp
for s in bench_targets bench_postproc bench_depthpost; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $OUT/${s}_$c -o p -- python3 $R/tools/$s.py --no-cpu > $OUT/${s}_$c.log 2>&1
  done
done
python3 - <<PY
import csv, glob, os, re
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        m = re.search(r"(?:\(anonymous namespace\)::)?(\w+)", k.replace("void ", ""))
        if not m or not re.match(r"(pt_|pp_|dp_)", m.group(1)):
            continue
        acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines = ["kernel            FETCH_SIZE_KB(avg)  WRITE_SIZE_KB(avg)  launches"]
for k, d in sorted(acc.items()):
    f, w = d.get("FETCH_SIZE", [0]), d.get("WRITE_SIZE", [0])
    lines.append(f"{k:16s} {sum(f)/len(f):18.1f} {sum(w)/len(w):19.1f} {len(f):9d}")
open("$OUT/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
