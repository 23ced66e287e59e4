#!/bin/bash
# round 3, call H: the whole GPU suite with the give-up path of the deferred tiles; long runs and chains; the generic engine at 256 MiB
O=gpurun_out/r3h; mkdir -p $O
timeout 500 python -m pytest tests -m gpu -x -q --timeout 200 -s 2>&1 | grep -v "^$" | tail -12 > $O/pytest_gpu.log; cat $O/pytest_gpu.log | cut -c1-300
timeout 200 python tools/stress_repeats.py o200k_shaped > $O/long_runs.txt 2>&1; cat $O/long_runs.txt | cut -c1-260
timeout 100 python tools/rx_diag.py > $O/rx_diag.txt 2>&1; grep encode $O/rx_diag.txt | cut -c1-300
timeout 150 python bench.py --generic-engine --mib 256 --steps 3 --warmup 1 --no-host-path --no-hf > $O/bench_generic_256.json 2> $O/gen.err
python - <<'PY'
import json
try:
    j=json.loads(open('gpurun_out/r3h/bench_generic_256.json').read().strip().splitlines()[-1]); km=j["roofline"]["kernels_ms_avg"]
    print("generic 256 MiB GB/s",j["value"],"parity",j["parity_all_tokens_vs_oracle"],{k:v for k,v in km.items() if 'rx' in k})
except Exception as e: print("failed",e)
PY
