#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf /tmp/pmcc
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/pmcc/a -o p -- python3 $R/tools/bench_one_conv.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_MFMA --output-format csv -d /tmp/pmcc/b -o p -- python3 $R/tools/bench_one_conv.py > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d /tmp/pmcc/c -o p -- python3 $R/tools/bench_one_conv.py > /dev/null 2>&1
python3 $R/tools/pmc_summary.py /tmp/pmcc conv
