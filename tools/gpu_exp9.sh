#!/bin/bash
O=gpurun_out/exp9; mkdir -p $O; : > $O/results.jsonl
run() {  # tag chunk_mib hwq
  GPU_MAX_HW_QUEUES=$3 TIKTOKEN_AMD_CHUNK_BYTES=$(( $2 << 20 )) timeout 200 python tools/exp_front.py --tag "$1" --no-parity 2>>$O/err.log | grep '^EXP ' | sed 's/^EXP //' >> $O/results.jsonl
  echo "$1 rc=$?"
}
run c1024 1024 4
run c512_q4 512 4
run c512_q12 512 12
run c256_q4 256 4
run c256_q12 256 12
run c128_q4 128 4
run c128_q12 128 12
python - <<'PY'
import json
for l in open('gpurun_out/exp9/results.jsonl'):
    j=json.loads(l)
    k=j["kernels_ms"]
    print(f'{j["tag"]:12s} {j["ms_per_step"]:7.3f} ms {j["gbps"]:7.1f} GB/s front {k.get("tk_k_front")} slow {k.get("tk_k_front_slow")} host {j["host"]}')
PY
