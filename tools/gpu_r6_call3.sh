#!/bin/bash
# Round 6, call 3: new phases E / F (class lists from the bitmap, dense claims) against round 5's form on one box; then the parity tests
mkdir -p gpurun_out
bash tools/gpu_ab.sh "ef1||" "ef0|ef0|" 2>&1 | tail -6
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -q -x 2>&1 | tail -8
