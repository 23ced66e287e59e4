#!/bin/bash
O=gpurun_out/exp8; mkdir -p $O; : > $O/results.jsonl
V=tiktoken_amd/csrc/variants
run() {  # tag lib extra-env
  env $3 TIKTOKEN_AMD_LIB=${2:+$PWD/$V/libtiktoken_amd_$2.so} timeout 200 python tools/exp_front.py --tag "$1" --no-parity 2>>$O/err.log | grep '^EXP ' | sed 's/^EXP //' >> $O/results.jsonl
  echo "$1 rc=$?"
}
run r02_a r02 A=1
run cur_a "" A=1
run cur_noprio "" TIKTOKEN_AMD_NOPRIO=1
run r02_b r02 A=1
run cur_b "" A=1
run cur_noprio_b "" TIKTOKEN_AMD_NOPRIO=1
python - <<'PY'
import json
for l in open('gpurun_out/exp8/results.jsonl'):
    j=json.loads(l)
    k=j["kernels_ms"]
    print(f'{j["tag"]:14s} {j["ms_per_step"]:7.3f} ms front {k.get("tk_k_front")} slow {k.get("tk_k_front_slow")} merges {sum(v for n,v in k.items() if "merge" in n):.3f} back {k.get("tk_k_back")} tf {k.get("tk_k_tile_finish")} docoff {k.get("tk_k_docoff")}')
PY
