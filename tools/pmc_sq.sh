#!/bin/bash
# usage: tools/pmc_sq.sh <tag> <bench args...>   -- SQ issue/stall counters + HBM traffic of one bench.py command, in
# separate rocprofv3 --pmc passes (8 SQ slots per pass; FETCH_SIZE and WRITE_SIZE cannot share a pass), then a summary.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
EXTRA="$*"
cd /tmp
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
run() { n=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/$n -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $EXTRA > $OUT/$n.log 2>&1; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run sq2 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
python3 $R/tools/pmc_summary.py $OUT ${FILTER:-reproj} > $OUT/summary.txt
cat $OUT/summary.txt
