#!/bin/bash
# usage: tools/pmc_traffic2.sh  -- calibrated HBM traffic of reproj_march<true> -> profiles/traffic.json (see tools/measure_traffic.py)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_traffic2
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-include-regex "reproj_march|reconstruct_kernel" --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- python3 $R/tools/measure_traffic.py run > $OUT/fetch.log 2>&1
rocprofv3 --kernel-include-regex "reproj_march|reconstruct_kernel" --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- python3 $R/tools/measure_traffic.py run > $OUT/write.log 2>&1
rocprofv3 --kernel-include-regex "reproj_march|reconstruct_kernel" --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/valu -o p -- python3 $R/tools/measure_traffic.py run > $OUT/valu.log 2>&1
python3 $R/tools/measure_traffic.py summarize $OUT | tee $OUT/summary.json
cp $R/profiles/traffic.json $R/gpurun_out/traffic.json
