#!/bin/bash
# usage: bash tools/gpu_round_end.sh TAG  -- everything the committed artefacts under profiles/ are made from:
# GPU tests, smoke, bench (1 GiB), kernel-trace stats + HBM traffic counters, SQ counters, the other configs, small calls, long runs,
# the generic pat_str engine beside the scanners, fuzzed batches.
TAG=${1:-r02}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/${TAG}_pytest_gpu.log; cat gpurun_out/${TAG}_pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 900 bash tools/gpu_prof.sh 1024 $TAG > gpurun_out/${TAG}_prof.log 2>&1; tail -3 gpurun_out/${TAG}_prof.log
timeout 900 bash tools/gpu_pmc.sh > gpurun_out/${TAG}_pmc.log 2>&1; tail -5 gpurun_out/${TAG}_pmc.log
timeout 600 python bench.py > gpurun_out/${TAG}_bench_1gpu.json 2> gpurun_out/${TAG}_bench.err; cat gpurun_out/${TAG}_bench_1gpu.json
timeout 600 python tools/bench_configs.py > gpurun_out/${TAG}_configs.jsonl 2> gpurun_out/${TAG}_configs.err; cat gpurun_out/${TAG}_configs.jsonl | cut -c1-300
timeout 300 python tools/small_call.py > gpurun_out/${TAG}_small_calls.txt 2>&1; cat gpurun_out/${TAG}_small_calls.txt
timeout 300 python tools/mid_call.py > gpurun_out/${TAG}_mid_calls.txt 2>&1; cat gpurun_out/${TAG}_mid_calls.txt
timeout 120 python tools/mid_corpus.py > gpurun_out/${TAG}_mid_calls_corpus.txt 2>&1; cat gpurun_out/${TAG}_mid_calls_corpus.txt
timeout 120 python tools/decode_path.py > gpurun_out/${TAG}_decode_path.txt 2>&1; cat gpurun_out/${TAG}_decode_path.txt
timeout 300 python tools/stress_repeats.py o200k_shaped > gpurun_out/${TAG}_long_runs.txt 2>&1; cat gpurun_out/${TAG}_long_runs.txt
timeout 300 python tools/generic_vs_scanners.py 256 > gpurun_out/${TAG}_generic_vs_scanners.txt 2>&1; cut -c1-400 gpurun_out/${TAG}_generic_vs_scanners.txt
timeout 120 python tools/rx_diag.py > gpurun_out/${TAG}_generic_pat_small_batches.txt 2>&1; tail -3 gpurun_out/${TAG}_generic_pat_small_batches.txt | cut -c1-300
timeout 300 python tools/gpu_fuzz.py 4 24 100 > gpurun_out/${TAG}_fuzz.txt 2>&1; tail -1 gpurun_out/${TAG}_fuzz.txt
for F in flat dfa program; do  # the generic engine's kernels in their three forms (the pattern's DFA with the one-loop speculative pass, piece by piece, the program)
  TIKTOKEN_AMD_RX_MATCHER=$F timeout 300 python bench.py --generic-engine --mib 256 --steps 3 --warmup 1 --no-host-path --no-hf --cpu-sample-mib 32 > gpurun_out/${TAG}_bench_generic_engine_256_$F.json 2> gpurun_out/${TAG}_bench_generic_engine_256_$F.err; cut -c100-300 gpurun_out/${TAG}_bench_generic_engine_256_$F.json
done
timeout 120 python tools/gpu_fuzz.py generic 60 777 > gpurun_out/${TAG}_fuzz_generic.txt 2>&1; tail -1 gpurun_out/${TAG}_fuzz_generic.txt
timeout 600 python bench.py --generic-engine --steps 3 --warmup 1 --no-host-path > gpurun_out/${TAG}_bench_generic_engine.json 2> gpurun_out/${TAG}_bench_generic_engine.err; cut -c1-300 gpurun_out/${TAG}_bench_generic_engine.json
