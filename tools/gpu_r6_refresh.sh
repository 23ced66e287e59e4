#!/bin/bash
# Round 6, after the last kernel change: the artefacts that depend on the kernel sources once more -- GPU tests, kernel stats + HBM traffic (the digest
# bench.py ties roofline.traffic to), SQ counters, the bench line, the configs, the generic engine at 256 MiB (in the bench line) and 1 GiB.
TAG=${1:-r06}
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 > gpurun_out/${TAG}_pytest_gpu.log; cat gpurun_out/${TAG}_pytest_gpu.log
timeout 600 bash tools/gpu_prof.sh 1024 $TAG > gpurun_out/${TAG}_prof.log 2>&1; tail -2 gpurun_out/${TAG}_prof.log | cut -c1-200
timeout 600 bash tools/gpu_pmc.sh > gpurun_out/${TAG}_pmc.log 2>&1; tail -3 gpurun_out/${TAG}_pmc.log
python tools/summarize_prof.py $TAG > /dev/null 2>&1; python tools/summarize_pmc.py $TAG > /dev/null 2>&1
timeout 500 python bench.py > gpurun_out/${TAG}_bench_1gpu.json 2> gpurun_out/${TAG}_bench.err; cut -c1-300 gpurun_out/${TAG}_bench_1gpu.json; tail -2 gpurun_out/${TAG}_bench.err
timeout 400 python tools/bench_configs.py > gpurun_out/${TAG}_configs.jsonl 2> gpurun_out/${TAG}_configs.err; cut -c1-200 gpurun_out/${TAG}_configs.jsonl
timeout 400 python bench.py --generic-engine --steps 3 --warmup 1 --no-host-path > gpurun_out/${TAG}_bench_generic_engine.json 2> gpurun_out/${TAG}_bench_generic_engine.err; cut -c1-300 gpurun_out/${TAG}_bench_generic_engine.json
cp profiles/${TAG}_kernel_stats.csv profiles/${TAG}_hbm_traffic.csv profiles/${TAG}_sq_counters.csv profiles/traffic.json gpurun_out/ 2>/dev/null
