#!/bin/bash
O=gpurun_out/exp3; mkdir -p $O; : > $O/results.jsonl
run() {  # tag chunk_mib hwq
  GPU_MAX_HW_QUEUES=$3 TIKTOKEN_AMD_CHUNK_BYTES=$(( $2 << 20 )) timeout 200 python tools/exp_front.py --tag "$1" --no-parity 2>>$O/err.log | grep '^EXP ' | sed 's/^EXP //' >> $O/results.jsonl
  echo "$1 rc=$?"
}
run c1024_q4 1024 4
run c1024_q16 1024 16
run c256_q8 256 8
run c256_q16 256 16
run c128_q8 128 8
run c128_q16 128 16
run c128_q24 128 24
run c64_q16 64 16
python - <<'PY'
import json
for l in open('gpurun_out/exp3/results.jsonl'):
    j=json.loads(l)
    k=j["kernels_ms"]
    print(f'{j["tag"]:12s} {j["ms_per_step"]:7.3f} ms {j["gbps"]:7.1f} GB/s sum_kernels {j["kernels_sum_ms"]} front {k.get("tk_k_front")} slow {k.get("tk_k_front_slow")} back {k.get("tk_k_back")} tf {k.get("tk_k_tile_finish")}')
PY
