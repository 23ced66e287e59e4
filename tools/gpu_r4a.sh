#!/bin/bash
# usage: bash tools/gpu_r4a.sh TAG -- GPU tests (stop at the first failure), bench, kernel stats + HBM traffic of the same command
TAG=${1:-r4a}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > gpurun_out/${TAG}_pytest_gpu.log; tail -15 gpurun_out/${TAG}_pytest_gpu.log
show() { python - "$1" <<'PY'
import json, sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], j["value"], j["ms_per_step"], j.get("parity_all_tokens_vs_oracle"))
    print({k:v for k,v in sorted(j["roofline"]["kernels_ms_avg"].items(), key=lambda x:-x[1])})
except Exception as e: print("no bench line", e)
PY
}
timeout 400 python bench.py > gpurun_out/${TAG}_bench_1gpu.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/${TAG}_bench.err
show gpurun_out/${TAG}_bench_1gpu.json
timeout 600 bash tools/gpu_prof.sh 1024 $TAG > gpurun_out/${TAG}_prof.log 2>&1; tail -25 gpurun_out/${TAG}_prof.log | cut -c1-200
