#!/bin/bash
# same-box A/B of stream / issue-order variants of the training step (plan replay, 40 timed steps each, alternating)
# usage: tools/ab_streams.sh "VAR=1 VAR2=1" ...   (one quoted env set per variant; "" = baseline)
out=gpurun_out/ab_streams.txt; : > $out
for rep in 1 2 3; do
  for v in "$@"; do
    r=$(env $v python bench.py --steps 40 --warmup 5 --exec plan --no-fp16-leg --no-cpu-baseline --no-host-probe 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['config']['step_execution'][:4])")
    echo "rep $rep [$v] $r" | tee -a $out
  done
done
