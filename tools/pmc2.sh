#!/bin/bash
# usage: tools/pmc2.sh <tag> <bench args...>  -- FETCH/WRITE/TCC + timing only
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
cd /tmp
mkdir -p $R/gpurun_out/pmc_$TAG
run() { n=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_$TAG/$n -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $EXTRA > $R/gpurun_out/pmc_$TAG/$n.log 2>&1; }
EXTRA="$*"
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline $EXTRA 2>&1 | tail -1 | grep -o "roofline.*"
