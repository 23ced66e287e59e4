#!/bin/bash
# Round 6: the GPU suite and the round's profiles on the final code -- tests (with the log of the reference-package / Appendix-B files), smoke, kernel stats + HBM
# traffic, SQ counters, the front kernel by phase (instruction counters and cycles per tile), bench line, configs, link rates, host path, the generic engine at 1 GiB.
TAG=${1:-r06}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 > gpurun_out/${TAG}_pytest_gpu.log; cat gpurun_out/${TAG}_pytest_gpu.log
timeout 300 python -m pytest tests/test_reference_package.py tests/test_real_vocab.py -m gpu -v 2>&1 | grep -v "^$" | tail -45 > gpurun_out/${TAG}_reference_package_over_shim.txt; tail -3 gpurun_out/${TAG}_reference_package_over_shim.txt
timeout 100 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 600 bash tools/gpu_prof.sh 1024 $TAG > gpurun_out/${TAG}_prof.log 2>&1; tail -2 gpurun_out/${TAG}_prof.log | cut -c1-200
timeout 600 bash tools/gpu_pmc.sh > gpurun_out/${TAG}_pmc.log 2>&1; tail -3 gpurun_out/${TAG}_pmc.log
timeout 600 bash tools/gpu_phases.sh $TAG > gpurun_out/${TAG}_front_phases.csv 2> gpurun_out/${TAG}_front_phases.err; cat gpurun_out/${TAG}_front_phases.csv
NOPAR=1 timeout 300 bash tools/gpu_ab.sh "timing|timing|" > gpurun_out/${TAG}_front_cycles.log 2>&1; python - <<PY > gpurun_out/${TAG}_front_cycles.txt
import json
for l in open("gpurun_out/ab/exp.jsonl"):
    j = json.loads(l); print(j["tag"], "front ms", j["kernels_ms"]["tk_k_front"], "cycles per tile and phase", j.get("cycles_per_tile"))
PY
cat gpurun_out/${TAG}_front_cycles.txt
timeout 100 tools/ubench/pcie_rates > gpurun_out/${TAG}_pcie_link.txt 2>&1; cat gpurun_out/${TAG}_pcie_link.txt
timeout 500 python bench.py > gpurun_out/${TAG}_bench_1gpu.json 2> gpurun_out/${TAG}_bench.err; cut -c1-400 gpurun_out/${TAG}_bench_1gpu.json; tail -2 gpurun_out/${TAG}_bench.err
timeout 400 python tools/bench_configs.py > gpurun_out/${TAG}_configs.jsonl 2> gpurun_out/${TAG}_configs.err; cut -c1-200 gpurun_out/${TAG}_configs.jsonl
timeout 300 python tools/small_call.py > gpurun_out/${TAG}_small_calls.txt 2>&1; tail -4 gpurun_out/${TAG}_small_calls.txt
timeout 120 python tools/mid_corpus.py > gpurun_out/${TAG}_mid_calls_corpus.txt 2>&1; tail -4 gpurun_out/${TAG}_mid_calls_corpus.txt
timeout 300 python tools/stress_repeats.py o200k_shaped > gpurun_out/${TAG}_long_runs.txt 2>&1; tail -5 gpurun_out/${TAG}_long_runs.txt
timeout 400 python bench.py --generic-engine --steps 3 --warmup 1 --no-host-path > gpurun_out/${TAG}_bench_generic_engine.json 2> gpurun_out/${TAG}_bench_generic_engine.err; cut -c1-300 gpurun_out/${TAG}_bench_generic_engine.json
