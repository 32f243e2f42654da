#!/bin/bash
# usage: tools/pmc_reproj.sh <tag> [env assignments...]  -- SQ / TA counters of reproj_march only (kernel filter: seconds per pass instead of minutes),
# loss-only bench (random depths / poses: incoherent gathers).  MGN_REPROJ_U8=1 selects the uint8 RGBX frame layout.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
run() { n=$1; shift; timeout 300 rocprofv3 --kernel-include-regex "reproj_march" --pmc "$@" --output-format csv -d $OUT/$n -o p -- python3 $R/bench.py --loss-only --steps 3 --warmup 1 --no-cpu-baseline > $OUT/$n.log 2>&1; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run sq2 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM
run ta1 TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum GRBM_GUI_ACTIVE
run ta2 TA_BUFFER_WAVEFRONTS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 $R/tools/pmc_summary.py $OUT reproj_march > $OUT/summary.txt
rm -rf $OUT/*/
