#!/bin/bash
# PMC counters of the semantic-head backward kernel (one rocprofv3 pass per counter group, each under its own timeout)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf /tmp/pmcu
pass() { d=$1; shift; timeout 150 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmcu/$d -o p -- python3 $R/tools/upce_one.py > $R/gpurun_out/pmc_upce_$d.log 2>&1; echo "pass $d rc=$?"; }
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT
pass b SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_MFMA
pass c SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR SQ_WAVES
python3 $R/tools/pmc_summary.py /tmp/pmcu upce_bwd
