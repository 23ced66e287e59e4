#!/bin/bash
# Round 3, call K: long pieces matched by the resolving wavefront together (tk_rx_match_dfa_coop) -- tests, adversarial batches, the 256 MiB line.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; O=gpurun_out/r3k
date +%s > ${O}_t0
( timeout 400 python -m pytest tests/test_gpu_regex.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider --timeout 150 -k "regex or generic or give_up or uncertain or ten_megabytes or fuzz or pattern or megabyte" 2>&1 | tail -30 ) > ${O}_pytest_gpu_rx.log; tail -3 ${O}_pytest_gpu_rx.log
timeout 120 python tools/rx_diag.py > ${O}_generic_pat_small_batches.txt 2>&1; grep -E "encode|pretokenize" ${O}_generic_pat_small_batches.txt | cut -c1-330
timeout 120 python tools/rx_diag.py "$(python -c "import sys; sys.path.insert(0,'tests'); import helpers as h; print(h.PAT_STR[2])")" runs,fuzz > ${O}_generic_o200k_small_batches.txt 2>&1; grep -E "encode" ${O}_generic_o200k_small_batches.txt | cut -c1-330
timeout 240 python bench.py --generic-engine --mib 256 --steps 3 --warmup 1 --no-host-path --no-hf --cpu-sample-mib 32 > ${O}_generic_flat.json 2> ${O}_generic_flat.err
python - <<PY
import json
try:
    d = json.load(open("${O}_generic_flat.json")); k = d["roofline"]["kernels_ms_avg"]
    print("flat", d["value"], "GB/s", d["ms_per_step"], "ms; parity", d["parity_all_tokens_vs_oracle"], {x: k[x] for x in k if "rx_" in x})
except Exception as e:
    print("flat: no line", e)
PY
timeout 200 python tools/stress_repeats.py o200k_shaped > ${O}_long_runs.txt 2>&1; tail -4 ${O}_long_runs.txt | cut -c1-200
timeout 300 python tools/gpu_fuzz.py generic 40 778 > ${O}_fuzz_generic.txt 2>&1; tail -1 ${O}_fuzz_generic.txt
timeout 300 python tools/gpu_fuzz.py 2 24 300 > ${O}_fuzz.txt 2>&1; tail -1 ${O}_fuzz.txt
echo "elapsed $(( $(date +%s) - $(cat ${O}_t0) )) s"
