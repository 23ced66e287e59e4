#!/bin/bash
# Round 3, call L: how far a speculative match may look beyond its segment (DFA forms), then the regex / long-piece GPU tests with the default.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; O=gpurun_out/r3l
date +%s > ${O}_t0
for A in 512 1024 2048 4096 16384; do
  echo "== TIKTOKEN_AMD_RX_AHEAD=$A" | tee -a ${O}_ahead_sweep.txt
  TIKTOKEN_AMD_RX_AHEAD=$A timeout 120 python tools/rx_diag.py '\w+|[^\w\s]+|\s+' runs,fuzz 2>&1 | grep -E "encode" | sed -n '2p;4p' | cut -c1-260 | tee -a ${O}_ahead_sweep.txt
done
for A in 1024 2048; do
  TIKTOKEN_AMD_RX_AHEAD=$A timeout 240 python bench.py --generic-engine --mib 256 --steps 3 --warmup 1 --no-host-path --no-cpu-baseline > ${O}_generic_a$A.json 2> ${O}_generic_a$A.err
  python - <<PY | tee -a ${O}_ahead_sweep.txt
import json
try:
    d = json.load(open("${O}_generic_a$A.json")); k = d["roofline"]["kernels_ms_avg"]
    print("ahead $A: 256 MiB web text", d["value"], "GB/s", d["ms_per_step"], "ms", {x: k[x] for x in k if "rx_" in x})
except Exception as e:
    print("ahead $A: no line", e)
PY
done
( timeout 400 python -m pytest tests/test_gpu_regex.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider --timeout 150 -k "regex or generic or give_up or uncertain or ten_megabytes or fuzz or pattern or megabyte" 2>&1 | tail -30 ) > ${O}_pytest_gpu_rx.log; tail -3 ${O}_pytest_gpu_rx.log
echo "elapsed $(( $(date +%s) - $(cat ${O}_t0) )) s"
