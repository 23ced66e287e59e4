#!/bin/bash
mkdir -p gpurun_out/s6
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -30 > gpurun_out/s6/pytest_gpu.log; grep -n "Error\|passed\|failed\|assert" gpurun_out/s6/pytest_gpu.log | head -20
timeout 300 python tools/small_call.py 2>&1 | tee gpurun_out/s6/small_call.txt
