#!/bin/bash
O=gpurun_out/exp7; mkdir -p $O; : > $O/results.jsonl
V=tiktoken_amd/csrc/variants
run() {  # tag lib
  TIKTOKEN_AMD_LIB=${2:+$PWD/$V/libtiktoken_amd_$2.so} timeout 200 python tools/exp_front.py --tag "$1" $3 2>>$O/err.log | grep '^EXP ' | sed 's/^EXP //' >> $O/results.jsonl
  echo "$1 rc=$?"
}
run rows ""
run norows norows
run r02 r02 --no-parity
python - <<'PY'
import json
for l in open('gpurun_out/exp7/results.jsonl'):
    j=json.loads(l)
    k=j["kernels_ms"]
    print(f'{j["tag"]:12s} {j["ms_per_step"]:7.3f} ms {j["gbps"]:7.1f} GB/s front {k.get("tk_k_front")} slow {k.get("tk_k_front_slow")} merges {sum(v for n,v in k.items() if "merge" in n):.3f} back {k.get("tk_k_back")} tf {k.get("tk_k_tile_finish")} parity {j.get("parity")}')
    print("   ", {n: v for n, v in k.items() if "merge" in n})
PY
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
