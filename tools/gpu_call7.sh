#!/bin/bash
# the staged speculative lanes with offsets, the start bits as LDS instructions, no clamp (new) against the committed form (st1), one box, interleaved; then the regex GPU tests
V=$PWD/tiktoken_amd/csrc/variants
run() {
  TIKTOKEN_AMD_LIB=${2:+$V/libtiktoken_amd_$2.so} timeout 200 python bench.py --generic-engine --mib 256 --steps 3 --warmup 1 --no-host-path --no-hf --cpu-sample-mib 32 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read()); k = j['roofline']['kernels_ms_avg']
print('$1 256 MiB: %.2f GB/s  %.3f ms  speculate %.3f link %.3f resolve %.3f merge %.3f front %.3f  parity %s' % (j['value'], j['ms_per_step'], k.get('tk_k_rx_speculate', 0), k.get('tk_k_rx_link', 0), k.get('tk_k_rx_resolve', 0), k.get('tk_k_rx_merge', 0), k.get('tk_k_front', 0), j['parity_all_tokens_vs_oracle']))
"
}
for rep in 1 2; do run committed st1; run new ""; done
timeout 300 python -m pytest tests/test_gpu_regex.py -m gpu -q -x 2>&1 | tail -2
TIKTOKEN_AMD_DEBUG=1048576 timeout 240 python tools/gpu_fuzz.py 1 16 301 2>&1 | tail -1
