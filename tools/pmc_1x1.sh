#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the streaming 1x1 kernel (tools/bench_1x1.py), separate passes
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_1x1
mkdir -p $OUT
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/$c -o p -- python3 $R/tools/bench_1x1.py > $OUT/$c.log 2>&1
done
python3 $R/tools/pmc_summary.py $OUT conv1x1_s | tee $OUT/summary.txt
python3 $R/tools/pmc_summary.py $OUT conv_wgrad_reduce | tee -a $OUT/summary.txt
