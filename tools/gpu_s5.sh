#!/bin/bash
mkdir -p gpurun_out/s5
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "runs or rounds or megabyte" 2>&1 | tail -30 > gpurun_out/s5/pytest_gpu.log; grep -n "Error\|passed\|failed\|assert" gpurun_out/s5/pytest_gpu.log | head -20
STRESS_SKIP_CHAINS=1 timeout 300 python tools/stress_repeats.py o200k_shaped 2>&1 > gpurun_out/s5/stress.txt; grep 1000000 gpurun_out/s5/stress.txt
