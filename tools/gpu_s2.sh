#!/bin/bash
# GPU session: parity tests first, then bench + per-phase breakdown of the new front kernel
mkdir -p gpurun_out/s2
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/s2/pytest_gpu.log; cat gpurun_out/s2/pytest_gpu.log
timeout 200 python bench.py --steps 5 --warmup 1 > gpurun_out/s2/bench.json 2> gpurun_out/s2/bench.err; echo "bench rc=$?"; cut -c1-1500 gpurun_out/s2/bench.json; tail -3 gpurun_out/s2/bench.err
PMC_VARIANTS="0 0x2000 0x4000 0x8000" bash tools/gpu_phases.sh r02b > gpurun_out/s2/phases.log 2>&1; tail -12 gpurun_out/s2/phases.log
