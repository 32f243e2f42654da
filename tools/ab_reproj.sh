#!/bin/bash
# quick loop for the reprojection kernel: parity tests + loss-only timing (+ optional A/B against another build of the lib)
python -m pytest tests/test_reproj_gpu.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do python bench.py --loss-only --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | grep -o '"avg_launch_ms": [0-9.]*'; done
if [ -n "$AB" ]; then MGNET_HIP_LIB=$PWD/$AB python bench.py --loss-only --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | grep -o '"avg_launch_ms": [0-9.]*'; fi
