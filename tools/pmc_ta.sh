#!/bin/bash
# usage: tools/pmc_ta.sh <tag> <bench args...>  -- vector-memory path counters (TA / TCP / TD) of one bench.py command
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
EXTRA="$*"
cd /tmp
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
run() { n=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/$n -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $EXTRA > $OUT/$n.log 2>&1; }
run ta1 TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum GRBM_GUI_ACTIVE
run ta2 TA_BUFFER_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
run tcp2 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum TD_TD_BUSY_sum
python3 $R/tools/pmc_summary.py $OUT ${FILTER:-reproj_march} > $OUT/summary.txt
cat $OUT/summary.txt
