#!/bin/bash
# front-kernel experiments, round 3 (run on the GPU box through gpurun): results -> gpurun_out/exp1/results.jsonl
O=gpurun_out/exp1; mkdir -p $O; : > $O/results.jsonl
V=tiktoken_amd/csrc/variants
run() {  # tag lib wgs dbg
  TIKTOKEN_AMD_LIB=${2:+$PWD/$V/libtiktoken_amd_$2.so} TIKTOKEN_AMD_FRONT_WGS=$3 TIKTOKEN_AMD_DEBUG=$4 timeout 200 python tools/exp_front.py --tag "$1" 2>>$O/err.log | grep '^EXP ' | sed 's/^EXP //' >> $O/results.jsonl
  echo "$1 rc=$?"
}
run r02 r02 "" ""
run default_hot10_w4 "" 4 ""
run hot10_w4_cacheoff "" 4 2097152
run hot0_w8 hot0_occ8 8 ""
run hot0_w6 hot0_occ8 6 ""
run hot0_w4 hot0_occ8 4 ""
run hot11_w3 hot11_occ3 3 ""
run hot9_w5 hot9_occ5 5 ""
run hot10_w3 "" 3 ""
python - <<'PY'
import json
for l in open('gpurun_out/exp1/results.jsonl'):
    j=json.loads(l)
    k=j["kernels_ms"]
    print(f'{j["tag"]:22s} {j["ms_per_step"]:7.3f} ms  front {k.get("tk_k_front")} slow {k.get("tk_k_front_slow")} back {k.get("tk_k_back")} tf {k.get("tk_k_tile_finish")} hit/probed {j.get("hot_hit_rate_of_probed")} hit/pieces {j.get("hot_hit_rate_of_pieces")} parity {j.get("parity")}')
PY
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
