#!/bin/bash
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -3
timeout 300 python bench.py --steps 5 --warmup 1 --no-host-path 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['roofline']['kernels_ms_avg']
print(j['value'], j['ms_per_step'], j['parity_all_tokens_vs_oracle'], {x:k[x] for x in k if 'merge' in x})"
