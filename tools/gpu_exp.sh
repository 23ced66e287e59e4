#!/bin/bash
# usage: bash tools/gpu_exp.sh "<dbg values>" [MIB]
mkdir -p gpurun_out
for D in $1; do
  TIKTOKEN_AMD_DEBUG=$D python bench.py --gpus 1 --steps 2 --warmup 1 --mib ${2:-1024} --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('dbg=$D', 'value', j['value'], 'ms', j['ms_per_step'], {k:v for k,v in r['kernels_ms_avg'].items() if v>0.08})
"
done
