#!/bin/bash
# round 6, call 6: workgroups per CU of the instance that finishes the deferred tiles (8 with spills, 6, 5), C3 and the other configs, one box
bash tools/gpu_ab.sh "g8|g8|" "g6|g6|" "g5|g5|"
V=$PWD/tiktoken_amd/csrc/variants
for o in 8 6 5 8 6 5; do echo "GIVEN_OCC=$o"; TIKTOKEN_AMD_LIB=$V/libtiktoken_amd_g$o.so python tools/bench_configs.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: j = json.loads(l)
    except Exception: continue
    if j['config'][:2] in ('C2', 'C5'): print(j['config'][:24], j['ms_per_step'], j.get('GBps'), round(j['kernels_ms'].get('tk_k_front_given', 0), 4))
"; done
