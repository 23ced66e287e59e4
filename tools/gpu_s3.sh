#!/bin/bash
mkdir -p gpurun_out/s3
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/s3/pytest_gpu.log; cat gpurun_out/s3/pytest_gpu.log
timeout 200 python bench.py --steps 5 --warmup 1 > gpurun_out/s3/bench.json 2> gpurun_out/s3/bench.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/s3/bench.json; python - <<'PY'
import json
j=json.loads(open('gpurun_out/s3/bench.json').read().strip().splitlines()[-1])
print({k:v for k,v in sorted(j["roofline"]["kernels_ms_avg"].items(), key=lambda x:-x[1])})
PY
timeout 300 python tools/stress_repeats.py o200k_shaped > gpurun_out/s3/stress.txt 2>&1; cat gpurun_out/s3/stress.txt
