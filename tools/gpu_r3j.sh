#!/bin/bash
# Round 3, last GPU call: the generic engine's DFA forms (tests, then the three forms side by side), the final bench line, kernel stats,
# small / adversarial batches, long runs, HBM traffic counters -- most important first, every step under its own timeout.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; O=gpurun_out/r3j
date +%s > ${O}_t0
( timeout 400 python -m pytest tests/test_gpu_regex.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider --timeout 150 -k "regex or generic or give_up or uncertain or ten_megabytes or fuzz or pattern" 2>&1 | tail -30 ) > ${O}_pytest_gpu_rx.log; tail -3 ${O}_pytest_gpu_rx.log
for F in flat dfa program; do
  TIKTOKEN_AMD_RX_MATCHER=$F timeout 240 python bench.py --generic-engine --mib 256 --steps 3 --warmup 1 --no-host-path --no-hf --cpu-sample-mib 32 > ${O}_generic_$F.json 2> ${O}_generic_$F.err
  python - <<PY
import json
try:
    d = json.load(open("${O}_generic_$F.json")); k = d["roofline"]["kernels_ms_avg"]
    print("$F", d["value"], "GB/s", d["ms_per_step"], "ms; parity", d["parity_all_tokens_vs_oracle"], {x: k[x] for x in k if "rx_" in x})
except Exception as e:
    print("$F: no line", e)
PY
done
timeout 400 python bench.py > ${O}_bench_1gpu.json 2> ${O}_bench.err; cut -c1-700 ${O}_bench_1gpu.json
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/${O}_trace -o r03 -- python $R/bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu-baseline --no-host-path > $R/${O}_trace.log 2>&1 )
f=$(find ${O}_trace -name '*kernel_stats.csv' | head -1); echo "== $f"; head -8 "$f" | cut -c1-200
find ${O}_trace -name '*.csv' -size +8M -delete
( timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 150 2>&1 | tail -30 ) > ${O}_pytest_gpu.log; tail -3 ${O}_pytest_gpu.log
timeout 120 python tools/rx_diag.py > ${O}_generic_pat_small_batches.txt 2>&1; tail -4 ${O}_generic_pat_small_batches.txt | cut -c1-400
TIKTOKEN_AMD_RX_MATCHER=program timeout 120 python tools/rx_diag.py '\w+|[^\w\s]+|\s+' fuzz > ${O}_generic_pat_small_batches_program.txt 2>&1; tail -2 ${O}_generic_pat_small_batches_program.txt | cut -c1-400
timeout 200 python tools/stress_repeats.py o200k_shaped > ${O}_long_runs.txt 2>&1; tail -12 ${O}_long_runs.txt
TIKTOKEN_AMD_RX_MATCHER=flat timeout 240 python bench.py --generic-engine --mib 1024 --steps 3 --warmup 1 --no-host-path --no-hf --cpu-sample-mib 32 > ${O}_generic_flat_1gib.json 2> ${O}_generic_flat_1gib.err; cut -c1-300 ${O}_generic_flat_1gib.json
for SH in 6 8 9; do
  TIKTOKEN_AMD_RX_SEG_SHIFT=$SH timeout 200 python bench.py --generic-engine --mib 256 --steps 3 --warmup 1 --no-host-path --no-cpu-baseline > ${O}_generic_flat_seg$SH.json 2> ${O}_generic_flat_seg$SH.err
  python -c "
import json
try:
    d = json.load(open('${O}_generic_flat_seg$SH.json')); k = d['roofline']['kernels_ms_avg']; print('seg shift $SH', d['value'], 'GB/s', {x: k[x] for x in k if 'rx_' in x})
except Exception as e: print('seg $SH: no line', e)
"
done
timeout 300 python tools/gpu_fuzz.py generic 60 777 > ${O}_fuzz_generic.txt 2>&1; tail -1 ${O}_fuzz_generic.txt
timeout 400 bash tools/gpu_prof.sh 1024 r3j > ${O}_pmc.log 2>&1; tail -8 ${O}_pmc.log
echo "elapsed $(( $(date +%s) - $(cat ${O}_t0) )) s"
