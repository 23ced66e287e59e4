#!/bin/bash
# usage: bash tools/gpu_ab.sh "tag|lib|env" ...   -- A/B runs of tools/exp_front.py on ONE box (boxes differ by a few percent), twice each, interleaved
mkdir -p gpurun_out/ab; : > gpurun_out/ab/exp.jsonl
V=$PWD/tiktoken_amd/csrc/variants
for rep in 1 2; do
for spec in "$@"; do
  IFS='|' read -r tag lib envs <<< "$spec"
  env $envs TIKTOKEN_AMD_LIB=${lib:+$V/libtiktoken_amd_$lib.so} timeout 300 python tools/exp_front.py --tag $tag ${NOPAR:+--no-parity} 2>>gpurun_out/ab/exp.err | grep '^EXP ' | sed 's/^EXP //' >> gpurun_out/ab/exp.jsonl
done; done
python - <<'PY'
import json
for l in open('gpurun_out/ab/exp.jsonl'):
    j=json.loads(l); k=j["kernels_ms"]
    print("%-14s %7.3f ch %s bs %s front %.3f slow %.3f+%.3f lists %.3f merges %.3f count %.3f scans %.3f place %.3f docoff %.3f parity %s" % (j["tag"], j["ms_per_step"], j["host"].get("chunks"), j["host"].get("back_streams"), k["tk_k_front"], k["tk_k_front_slow"], k.get("tk_k_front_given",0), k.get("tk_k_bincount",0)+k.get("tk_k_binfill",0), sum(v for n,v in k.items() if "merge" in n), k.get("tk_k_count_tiles",0), sum(v for n,v in k.items() if "scan" in n), k["tk_k_place"], k.get("tk_k_docoff", 0), j.get("parity")))
PY
