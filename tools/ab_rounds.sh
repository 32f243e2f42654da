#!/bin/bash
# same-box comparison of two trees: the round-4 tree (exported to ab_r4/ with `git archive 65c67cb`, built there) against this tree;
# alternating runs of the default benchmark (50 timed steps, plan replay and the auto calibration's pick), one JSON summary line per run
out=gpurun_out/ab_rounds.txt; : > $out
for rep in 1 2 3 4; do
  for tree in ab_r4 .; do
    r=$(cd $tree && python bench.py --steps 50 --warmup 10 --no-fp16-leg --no-cpu-baseline --no-host-probe 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); p=d['config'].get('step_execution_probe',{}); print(d['ms_per_step'], d['value'], d['config']['step_execution'][:5], 'probe plan/eager', p.get('plan_ms_per_step'), p.get('eager_ms_per_step'))")
    echo "rep $rep tree $tree: $r" | tee -a $out
  done
done
