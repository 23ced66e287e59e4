#!/bin/bash
mkdir -p gpurun_out/s4
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/s4/pytest_gpu.log; cat gpurun_out/s4/pytest_gpu.log
timeout 200 python tools/debug_t2.py 512 2>&1 | tail -12
timeout 300 python bench.py --steps 5 --warmup 1 > gpurun_out/s4/bench.json 2> gpurun_out/s4/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/s4/bench.err; python - <<'PY'
import json
j=json.loads(open('gpurun_out/s4/bench.json').read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], j["parity_all_tokens_vs_oracle"], j["host_path"], j["cpu_baseline"]["value"])
print({k:v for k,v in sorted(j["roofline"]["kernels_ms_avg"].items(), key=lambda x:-x[1])})
PY
