#!/bin/bash
# PMC counters of conv3x3_up2_win (one rocprofv3 pass per counter group); usage: tools/pmc_up2.sh <tag>
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1
cd /tmp
rm -rf /tmp/pmcu_$TAG
pass() { d=$1; shift; timeout 150 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmcu_$TAG/$d -o p -- python3 $R/tools/up2_one.py > $R/gpurun_out/pmc_up2_${TAG}_$d.log 2>&1; echo "pass $d rc=$?"; }
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT
pass b SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_MFMA
python3 $R/tools/pmc_summary.py /tmp/pmcu_$TAG conv3x3_up2
