#!/bin/bash
# Round 3, call M (the last): the final build -- whole GPU suite, the bench line, the generic engine at 256 MiB and 1 GiB, small / adversarial batches.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; O=gpurun_out/r3m
date +%s > ${O}_t0
( timeout 200 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 120 2>&1 | tail -30 ) > ${O}_pytest_gpu.log; tail -3 ${O}_pytest_gpu.log
timeout 100 python bench.py > ${O}_bench_1gpu.json 2> ${O}_bench.err; cut -c1-330 ${O}_bench_1gpu.json
timeout 60 python bench.py --generic-engine --mib 256 --steps 3 --warmup 1 --no-host-path --no-hf --cpu-sample-mib 32 > ${O}_generic_256.json 2> ${O}_generic_256.err; cut -c100-330 ${O}_generic_256.json
timeout 60 python tools/rx_diag.py > ${O}_generic_pat_small_batches.txt 2>&1; grep -E "encode" ${O}_generic_pat_small_batches.txt | cut -c1-200
timeout 80 python bench.py --generic-engine --mib 1024 --steps 3 --warmup 1 --no-host-path --no-hf --cpu-sample-mib 32 > ${O}_generic_1gib.json 2> ${O}_generic_1gib.err; cut -c100-330 ${O}_generic_1gib.json
echo "elapsed $(( $(date +%s) - $(cat ${O}_t0) )) s"
