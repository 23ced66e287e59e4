#!/bin/bash
for E in gpt2_shaped cl100k_shaped o200k_shaped; do
  python bench.py --gpus 1 --steps 2 --warmup 1 --mib 1024 --no-cpu-baseline --encoding $E 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('$E', 'value', j['value'], 'ms', j['ms_per_step'], 'pieces', j['config']['pieces_rank0'], {k:v for k,v in r['kernels_ms_avg'].items() if v>0.4})
"
done
