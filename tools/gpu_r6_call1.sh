#!/bin/bash
# Round 6, first GPU call: the GPU suite on the round's first commit (new tests included), the N > 1 dry run with the two-phase gather probe,
# the baseline bench line before any kernel change, N1g.
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > gpurun_out/r06a_pytest_gpu.log; cat gpurun_out/r06a_pytest_gpu.log
timeout 300 python -m pytest tests/test_reference_package.py tests/test_real_vocab.py -m gpu -v 2>&1 | tail -40 > gpurun_out/r06a_refpkg.log; tail -25 gpurun_out/r06a_refpkg.log
timeout 400 bash tools/gpu_bench_n2_dry.sh 256 > gpurun_out/r06a_bench_n2_dry.txt 2>&1; tail -c 1500 gpurun_out/r06a_bench_n2_dry.txt
timeout 500 python bench.py > gpurun_out/r06a_bench_1gpu.json 2> gpurun_out/r06a_bench.err; cut -c1-600 gpurun_out/r06a_bench_1gpu.json; tail -3 gpurun_out/r06a_bench.err
