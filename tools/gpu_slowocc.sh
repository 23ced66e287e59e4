#!/bin/bash
V=$PWD/tiktoken_amd/csrc/variants
for rep in 1 2; do for lib in "" so3; do
  echo "== lib=${lib:-base}"; TIKTOKEN_AMD_LIB=${lib:+$V/libtiktoken_amd_$lib.so} timeout 300 python tools/bench_configs.py C2 C5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    j=json.loads(l); k=j['kernels_ms_avg']; print(j['config'][:3], j['ms_per_step'], j['GBps'], j['parity_all_tokens'], 'slow', k.get('tk_k_front_slow'), 'front', k.get('tk_k_front'))
"; done; done
NOPAR=1 bash tools/gpu_ab.sh "base||" "so3|so3|" 2>&1 | tail -4
