#!/bin/bash
# round 2, GPU session: instruction-rate microbenchmark, baseline bench, per-phase breakdown of tk_k_front (every step under its own timeout)
mkdir -p gpurun_out/s1
timeout 120 ./tools/ubench/valu_rates > gpurun_out/s1/valu_rates.txt 2>&1; echo "ubench rc=$?"; cat gpurun_out/s1/valu_rates.txt
timeout 200 python bench.py --steps 5 --warmup 1 > gpurun_out/s1/bench_base.json 2> gpurun_out/s1/bench_base.err; echo "bench rc=$?"; cut -c1-1200 gpurun_out/s1/bench_base.json; tail -3 gpurun_out/s1/bench_base.err
bash tools/gpu_phases.sh r02a > gpurun_out/s1/phases.log 2>&1; tail -12 gpurun_out/s1/phases.log
