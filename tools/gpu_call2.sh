#!/bin/bash
# Round 5, GPU call 2 of the second session: the generic engine's speculative pass over text staged in LDS (tk_k_rx_speculate_staged) --
# GPU tests, the one-loop lanes ($TIKTOKEN_AMD_RX_STAGED=0) against it on one box, fuzzed batches forced through the generic engine, its bench lines.
TAG=r05
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 > gpurun_out/${TAG}_pytest_gpu_2.log; cat gpurun_out/${TAG}_pytest_gpu_2.log
O=gpurun_out/${TAG}_generic_staged.txt; : > $O
for rep in 1 2; do for st in 0 1; do
  TIKTOKEN_AMD_RX_STAGED=$st timeout 200 python bench.py --generic-engine --mib 256 --steps 3 --warmup 1 --no-host-path --no-hf --cpu-sample-mib 32 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read()); k = j['roofline']['kernels_ms_avg']
print('staged=$st 256 MiB: %.2f GB/s  %.3f ms  speculate %.3f link %.3f resolve %.3f merge %.3f front %.3f  parity %s' % (j['value'], j['ms_per_step'], k.get('tk_k_rx_speculate', 0), k.get('tk_k_rx_link', 0), k.get('tk_k_rx_resolve', 0), k.get('tk_k_rx_merge', 0), k.get('tk_k_front', 0), j['parity_all_tokens_vs_oracle']))
" >> $O
done; done
cat $O
TIKTOKEN_AMD_DEBUG=1048576 timeout 240 python tools/gpu_fuzz.py 1 16 300 2>&1 | tail -2 | tee -a $O
timeout 120 python tools/gpu_fuzz.py generic 40 4242 2>&1 | tail -1 | tee -a $O
timeout 200 python tools/generic_vs_scanners.py 256 2>&1 | cut -c1-400 | tee -a $O
timeout 400 python bench.py --generic-engine --steps 3 --warmup 1 --no-host-path > gpurun_out/${TAG}_bench_generic_engine.json 2> gpurun_out/${TAG}_bench_generic_engine.err; cut -c1-300 gpurun_out/${TAG}_bench_generic_engine.json
