#!/bin/bash
# usage: bash tools/gpu_round_end.sh TAG  -- everything the committed artefacts under profiles/ are made from:
# GPU tests, smoke, bench (1 GiB), kernel-trace stats + HBM traffic counters, SQ counters, the other configs.
TAG=${1:-r01}
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/${TAG}_pytest_gpu.log; cat gpurun_out/${TAG}_pytest_gpu.log
python __graft_entry__.py smoke 2>&1 | tail -1
bash tools/gpu_prof.sh 1024 $TAG > gpurun_out/${TAG}_prof.log 2>&1; tail -3 gpurun_out/${TAG}_prof.log
bash tools/gpu_pmc.sh > gpurun_out/${TAG}_pmc.log 2>&1; tail -40 gpurun_out/${TAG}_pmc.log
python bench.py > gpurun_out/${TAG}_bench_1gpu.json 2> gpurun_out/${TAG}_bench.err; cat gpurun_out/${TAG}_bench_1gpu.json
python tools/bench_configs.py > gpurun_out/${TAG}_configs.jsonl 2> gpurun_out/${TAG}_configs.err; cat gpurun_out/${TAG}_configs.jsonl | cut -c1-300
