#!/bin/bash
O=gpurun_out/exp4; mkdir -p $O; : > $O/results.jsonl
run() {  # tag chunk_mib
  TIKTOKEN_AMD_CHUNK_BYTES=$(( $2 << 20 )) timeout 200 python tools/exp_front.py --tag "$1" $3 2>>$O/err.log | grep '^EXP ' | sed 's/^EXP //' >> $O/results.jsonl
  echo "$1 rc=$?"
}
run chunk1024 1024
run chunk512 512 --no-parity
run chunk256 256
run chunk128 128
run chunk64 64 --no-parity
python - <<'PY'
import json
for l in open('gpurun_out/exp4/results.jsonl'):
    j=json.loads(l)
    k=j["kernels_ms"]
    print(f'{j["tag"]:12s} {j["ms_per_step"]:7.3f} ms {j["gbps"]:7.1f} GB/s sum_kernels {j["kernels_sum_ms"]} front {k.get("tk_k_front")} slow {k.get("tk_k_front_slow")} back {k.get("tk_k_back")} tf {k.get("tk_k_tile_finish")} parity {j.get("parity")}')
PY
bash tools/gpu_timeline.sh 128 c128b > /dev/null 2>&1
bash tools/gpu_timeline.sh 256 c256b > /dev/null 2>&1
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
