#!/bin/bash
# PMC counters of EVERY kernel of two benchmark steps (two rocprofv3 passes, each under its own timeout) -> per-kernel table
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf /tmp/pmcstep
pass() { d=$1; shift; timeout 240 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmcstep/$d -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-host-probe --no-cpu-baseline > $R/gpurun_out/pmc_step_$d.log 2>&1; echo "pass $d rc=$?"; }
pass a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES
python3 $R/tools/pmc_step_table.py /tmp/pmcstep
