#!/bin/bash
# Round 3, call P (the last seconds): the generic engine's lines with the final tables (start states by the previous char's group).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; O=gpurun_out/r3p
for M in 256 256 1024; do
  timeout 25 python bench.py --generic-engine --mib $M --steps 3 --warmup 1 --no-host-path --no-cpu-baseline 2>/dev/null | tee ${O}_generic_$M.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print($M, d['value'], d['ms_per_step'], {k:v for k,v in d['roofline']['kernels_ms_avg'].items() if 'rx_' in k})"
done
