#!/bin/bash
# usage: tools/pmc.sh <tag>   -- collects PMC passes for bench.py (short run) into gpurun_out/pmc_<tag>/
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1
cd /tmp
run() { # name counters...
  n=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_$TAG/$n -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_$TAG/$n.log 2>&1
}
mkdir -p $R/gpurun_out/pmc_$TAG
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run grbm GRBM_GUI_ACTIVE
ls -R $R/gpurun_out/pmc_$TAG | head -30
