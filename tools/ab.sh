timeout 600 python -m pytest tests/test_headloss_gpu.py tests/test_network_gpu.py tests/test_model_golden.py -q 2>&1 | grep -E "^E|passed|failed|Error" | head
for i in 1 2; do python bench.py --no-cpu-baseline --steps 15 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*'; done
