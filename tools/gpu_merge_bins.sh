#!/bin/bash
# usage: bash tools/gpu_merge_bins.sh [DEBUG VALUES...] -- tk_k_merge_all with only ONE length bin merged (debug bits 25..28 = bin + 1; wrong tokens: timing only);
# further bits: 0x1000000 no probes, 0x80000 one merge per step
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/mbins; : > gpurun_out/mbins/out.txt
for D in "$@"; do
  TIKTOKEN_AMD_DEBUG=$((D)) timeout 120 python tools/exp_front.py --tag d$D --steps 2 --no-parity 2>/dev/null | grep '^EXP ' | sed 's/^EXP //' | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('debug', '$D', 'merge_all ms', j['kernels_ms'].get('tk_k_merge_all'), 'step', j['ms_per_step'])" >> gpurun_out/mbins/out.txt
done
cat gpurun_out/mbins/out.txt
