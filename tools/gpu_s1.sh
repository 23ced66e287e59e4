#!/bin/bash
# round 2, GPU session 1: instruction-rate microbenchmark, baseline bench, per-phase breakdown of tk_k_front
mkdir -p gpurun_out/s1
./tools/ubench/valu_rates > gpurun_out/s1/valu_rates.txt 2>&1; cat gpurun_out/s1/valu_rates.txt
python bench.py --steps 5 --warmup 1 > gpurun_out/s1/bench_base.json 2> gpurun_out/s1/bench_base.err; cat gpurun_out/s1/bench_base.json | cut -c1-1500
bash tools/gpu_phases.sh r02a > gpurun_out/s1/phases.log 2>&1; tail -12 gpurun_out/s1/phases.log
