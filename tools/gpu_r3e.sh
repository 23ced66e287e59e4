#!/bin/bash
# round 3, call E: the GPU suite (per-test time limit), the bench line, rocprofv3 kernel stats of the same command, mid-size calls, the generic engine
O=gpurun_out/r3e; mkdir -p $O
timeout 420 python -m pytest tests -m gpu -x -q --timeout 150 2>&1 | tail -6 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
timeout 400 python bench.py > $O/bench_1gpu.json 2> $O/bench.err; cut -c1-600 $O/bench_1gpu.json; tail -2 $O/bench.err
R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o r03 -- python $R/bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu-baseline --no-host-path > $R/$O/trace.log 2>&1 )
f=$(find $O/trace -name '*kernel_stats.csv' | head -1); echo "== $f"; head -12 "$f"
find $O/trace -name '*kernel_trace.csv' -size +8M -delete; find $O/trace -name '*.csv' -size +8M -delete
timeout 120 python tools/mid_call.py > $O/mid_calls.txt 2>&1; cat $O/mid_calls.txt | cut -c1-400
timeout 200 python bench.py --generic-engine --mib 256 --steps 3 --warmup 1 --no-host-path --no-hf > $O/bench_generic_256.json 2> $O/bench_generic.err; cut -c1-300 $O/bench_generic_256.json; python - <<'PY'
import json
try:
    j=json.loads(open('gpurun_out/r3e/bench_generic_256.json').read().strip().splitlines()[-1]); print(j["value"], j["parity_all_tokens_vs_oracle"], j["roofline"]["kernels_ms_avg"])
except Exception as e: print("generic:", e)
PY
