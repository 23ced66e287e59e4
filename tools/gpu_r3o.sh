#!/bin/bash
# Round 3, call O: tables for patterns that look at the char before the position (\b, one-char look-behind): the regex / long-piece GPU tests, the 256 MiB line.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; O=gpurun_out/r3o
( timeout 90 python -m pytest tests/test_gpu_regex.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider --timeout 60 -k "regex or generic or give_up or uncertain or ten_megabytes or fuzz or pattern or megabyte" 2>&1 | tail -30 ) > ${O}_pytest_gpu_rx.log; tail -3 ${O}_pytest_gpu_rx.log
timeout 40 python bench.py --generic-engine --mib 256 --steps 3 --warmup 1 --no-host-path --no-hf --cpu-sample-mib 32 > ${O}_generic_256.json 2> ${O}_generic_256.err; cut -c100-260 ${O}_generic_256.json; python -c "
import json; d=json.load(open('${O}_generic_256.json')); print(d['parity_all_tokens_vs_oracle'], {k:v for k,v in d['roofline']['kernels_ms_avg'].items() if 'rx_' in k})"
timeout 30 python tools/rx_diag.py '\b\w+\b|\s+|\B[^\w\s]+|[^\w\s]' words,fuzz > ${O}_wordb.txt 2>&1; grep encode ${O}_wordb.txt | cut -c1-200
