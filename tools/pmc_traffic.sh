#!/bin/bash
# usage: tools/pmc_traffic.sh <tag>  -- FETCH_SIZE / WRITE_SIZE (+ TCC miss) passes of bench.py for the roofline `traffic` field
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum --output-format csv -d $OUT/write -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/write.log 2>&1
python3 $R/tools/pmc_summary.py $OUT reproj_march | tee $OUT/summary_reproj.txt
python3 $R/tools/pmc_summary.py $OUT conv_igemm_big256 | tee $OUT/summary_big256.txt
