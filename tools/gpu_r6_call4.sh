#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > gpurun_out/r06c_pytest_gpu.log; cat gpurun_out/r06c_pytest_gpu.log
bash tools/gpu_pmc_ab.sh "ef3||" 2>&1 | grep "front<2, false, 0>"
timeout 300 python tools/bench_configs.py C1 C2 C5 N1 2>/dev/null | cut -c1-260
