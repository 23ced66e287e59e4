#!/bin/bash
mkdir -p gpurun_out/s8
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -30 > gpurun_out/s8/pytest_gpu.log; grep -n "Error\|passed\|failed\|assert" gpurun_out/s8/pytest_gpu.log | head -30
timeout 300 python bench.py --steps 5 --warmup 1 --no-host-path > gpurun_out/s8/bench.json 2> gpurun_out/s8/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
j=json.loads(open('gpurun_out/s8/bench.json').read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], j["parity_all_tokens_vs_oracle"])
print({k:v for k,v in sorted(j["roofline"]["kernels_ms_avg"].items(), key=lambda x:-x[1])})
PY
timeout 200 python tools/stress_repeats.py o200k_shaped 2>&1 | grep "Ab\|x'll" 
