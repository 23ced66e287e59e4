#!/bin/bash
# chunk pipelining experiments: results -> gpurun_out/exp2/results.jsonl
O=gpurun_out/exp2; mkdir -p $O; : > $O/results.jsonl
run() {  # tag chunk_mib
  TIKTOKEN_AMD_CHUNK_BYTES=$(( $2 << 20 )) timeout 200 python tools/exp_front.py --tag "$1" 2>>$O/err.log | grep '^EXP ' | sed 's/^EXP //' >> $O/results.jsonl
  echo "$1 rc=$?"
}
run chunk1024 1024
run chunk512 512
run chunk256 256
run chunk128 128
run chunk64 64
run chunk32 32
python - <<'PY'
import json
for l in open('gpurun_out/exp2/results.jsonl'):
    j=json.loads(l)
    k=j["kernels_ms"]
    print(f'{j["tag"]:12s} {j["ms_per_step"]:7.3f} ms {j["gbps"]:7.1f} GB/s sum_kernels {j["kernels_sum_ms"]} front {k.get("tk_k_front")} slow {k.get("tk_k_front_slow")} back {k.get("tk_k_back")} tf {k.get("tk_k_tile_finish")} parity {j.get("parity")}')
PY
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
