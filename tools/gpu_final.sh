#!/bin/bash
# Round 5, last GPU call: the GPU suite and the round's profiles on the final code (kernel stats, HBM traffic, SQ counters, bench line, configs, the generic engine at 1 GiB)
TAG=r05
mkdir -p gpurun_out
timeout 700 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 > gpurun_out/${TAG}_pytest_gpu.log; cat gpurun_out/${TAG}_pytest_gpu.log
timeout 100 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 600 bash tools/gpu_prof.sh 1024 $TAG > gpurun_out/${TAG}_prof.log 2>&1; tail -2 gpurun_out/${TAG}_prof.log | cut -c1-200
timeout 600 bash tools/gpu_pmc.sh > gpurun_out/${TAG}_pmc.log 2>&1; tail -3 gpurun_out/${TAG}_pmc.log
timeout 400 python bench.py > gpurun_out/${TAG}_bench_1gpu.json 2> gpurun_out/${TAG}_bench.err; cut -c1-400 gpurun_out/${TAG}_bench_1gpu.json
timeout 300 python tools/bench_configs.py > gpurun_out/${TAG}_configs.jsonl 2> gpurun_out/${TAG}_configs.err; cut -c1-200 gpurun_out/${TAG}_configs.jsonl
timeout 400 python bench.py --generic-engine --steps 3 --warmup 1 --no-host-path > gpurun_out/${TAG}_bench_generic_engine.json 2> gpurun_out/${TAG}_bench_generic_engine.err; cut -c1-300 gpurun_out/${TAG}_bench_generic_engine.json
