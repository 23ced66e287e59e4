#!/bin/bash
# round 3, call I: the GPU suite and the bench line with the decode rate and the native marshalling
O=gpurun_out/r3i; mkdir -p $O
timeout 500 python -m pytest tests -m gpu -x -q --timeout 200 2>&1 | tail -5 > $O/pytest_gpu.log; cat $O/pytest_gpu.log | cut -c1-300
timeout 400 python bench.py > $O/bench_1gpu.json 2> $O/bench.err; tail -2 $O/bench.err; python - <<'PY'
import json
j=json.loads(open('gpurun_out/r3i/bench_1gpu.json').read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], j["parity_all_tokens_vs_oracle"], json.dumps(j["host_path"])[:1200])
PY
