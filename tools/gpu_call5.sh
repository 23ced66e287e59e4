#!/bin/bash
# duplicates that look again at a slot whose claimant has not written yet (TKF_CLAIM_SPIN 0 = shipped / 8 / 32): C1, C2, C3 on one box
V=$PWD/tiktoken_amd/csrc/variants
for rep in 1 2; do for v in "" spin8 spin32; do
  echo "== ${v:-spin0}"
  TIKTOKEN_AMD_LIB=${v:+$V/libtiktoken_amd_$v.so} timeout 100 python tools/bench_configs.py C1 C2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); k = j['kernels_ms_avg']
    print('%s %.3f ms per step, front %.4f merge %.4f place %.4f  parity %s' % (j['config'][:2], j['ms_per_step'], k.get('tk_k_front', 0), k.get('tk_k_merge_all', 0), k.get('tk_k_place', 0), j['parity_all_tokens']))
"
done; done
bash tools/gpu_ab.sh "spin0||" "spin8|spin8|" "spin32|spin32|"
