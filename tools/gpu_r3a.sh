#!/bin/bash
# round 3, call A: state of HEAD -- GPU tests, bench line, per-phase times of the front kernel (no PMC passes)
mkdir -p gpurun_out/r3a
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r3a/pytest_gpu.log; cat gpurun_out/r3a/pytest_gpu.log
timeout 600 python bench.py > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err; cut -c1-1500 gpurun_out/r3a/bench.json; tail -3 gpurun_out/r3a/bench.err
PMC_VARIANTS=" " timeout 600 bash tools/gpu_phases.sh r3a 2>&1 | tail -12
