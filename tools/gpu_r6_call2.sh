#!/bin/bash
# Round 6, second GPU call: PC sampling of the pipeline (what the front kernel's instruction diet is steered by), the N > 1 dry run again with its exit code
mkdir -p gpurun_out
timeout 500 bash tools/gpu_pcsample.sh stochastic 1048576 1024 2>&1 | tail -25
ls -la gpurun_out/pcs/
if [ ! -s gpurun_out/pcs/hist_stochastic_by_pc.csv ]; then timeout 500 bash tools/gpu_pcsample.sh host_trap 256 1024 2>&1 | tail -25; fi
timeout 300 bash tools/gpu_bench_n2_dry.sh 128 > gpurun_out/r06b_bench_n2_dry.txt 2>&1; tail -c 1200 gpurun_out/r06b_bench_n2_dry.txt
