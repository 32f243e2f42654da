# fp16 leg of the default bench run vs standalone fp16 / bf16 runs on one box (ms per step [, fp16 leg])
cd $GRAFT_REPO_ROOT
g() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], (d.get('fp16') or {}).get('ms_per_step'))"; }
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | g default_with_leg
timeout 200 python bench.py --no-cpu-baseline --no-fp16-leg --dtype fp16 2>/dev/null | g fp16_alone
timeout 200 python bench.py --no-cpu-baseline --no-fp16-leg 2>/dev/null | g bf16_alone
timeout 200 python bench.py --no-cpu-baseline --no-fp16-leg --dtype fp16 2>/dev/null | g fp16_alone
timeout 200 python bench.py --no-cpu-baseline --no-fp16-leg 2>/dev/null | g bf16_alone
