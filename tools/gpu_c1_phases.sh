#!/bin/bash
# C1 (1 MiB of Lorem ipsum, one document, gpt2-shaped): the front kernel stopped after each phase (debug bits 0x1000 .. 0x10000), without probes (2), without claims (8), without the in-call table (256)
for D in 0 4096 8192 16384 32768 65536 2 8 256; do
  TIKTOKEN_AMD_DEBUG=$D timeout 100 python tools/bench_configs.py C1 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); k = j['kernels_ms_avg']
    print('dbg %6d: %.3f ms per step, front %.4f place %.4f merge %.4f count %.4f scan %.4f  sum %.3f  parity %s' % ($D, j['ms_per_step'], k.get('tk_k_front', 0), k.get('tk_k_place', 0), k.get('tk_k_merge_all', 0), k.get('tk_k_count_tiles', 0), k.get('tk_k_scan_small', 0), sum(k.values()), j['parity_all_tokens']))
"
done
