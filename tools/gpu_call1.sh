#!/bin/bash
# Round 5, GPU call 1 of the second session: GPU tests of the three-instance front kernel, TKF_SLOW_OCC 4 / 3 / 2 against the committed
# two-instance form on one box, C2 / C5 per variant, then the round's profiles (kernel stats, HBM traffic, SQ counters, bench line, configs) on the winner.
TAG=r05
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 > gpurun_out/${TAG}_pytest_gpu.log; cat gpurun_out/${TAG}_pytest_gpu.log
timeout 100 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 500 bash tools/gpu_ab.sh "new||" "base|base|" "occ2|occ2|" "occ3|occ3|" > gpurun_out/${TAG}_slow_occ_ab.txt 2>&1; cat gpurun_out/${TAG}_slow_occ_ab.txt
V=$PWD/tiktoken_amd/csrc/variants
for v in "" occ2 occ3; do
  echo "== config C2, variant '${v:-new}'" >> gpurun_out/${TAG}_slow_occ_ab.txt
  TIKTOKEN_AMD_LIB=${v:+$V/libtiktoken_amd_$v.so} timeout 200 python tools/bench_configs.py C2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); k = j['kernels_ms_avg']
    print(j['config'][:2], j['GBps'], 'GB/s', j['ms_per_step'], 'ms  front', k.get('tk_k_front'), 'slow', k.get('tk_k_front_slow'), 'given', k.get('tk_k_front_given'), 'parity', j['parity_all_tokens'])
" >> gpurun_out/${TAG}_slow_occ_ab.txt
done
tail -12 gpurun_out/${TAG}_slow_occ_ab.txt
# the winner among the three-instance builds (mean ms per step at C3)
BEST=$(python - <<'PY'
import json, collections
d = collections.defaultdict(list)
for l in open('gpurun_out/ab/exp.jsonl'):
    j = json.loads(l)
    if j['tag'] != 'base' and j.get('parity', True):
        d[j['tag']].append(j['ms_per_step'])
best = min(d, key=lambda t: sum(d[t]) / len(d[t])) if d else 'new'
print('' if best == 'new' else best)
PY
)
echo "profiles with variant '${BEST:-new}'" | tee gpurun_out/${TAG}_profile_variant.txt
export TIKTOKEN_AMD_LIB=${BEST:+$V/libtiktoken_amd_$BEST.so}
timeout 600 bash tools/gpu_prof.sh 1024 $TAG > gpurun_out/${TAG}_prof.log 2>&1; tail -3 gpurun_out/${TAG}_prof.log
timeout 600 bash tools/gpu_pmc.sh > gpurun_out/${TAG}_pmc.log 2>&1; tail -5 gpurun_out/${TAG}_pmc.log
timeout 400 python bench.py > gpurun_out/${TAG}_bench_1gpu.json 2> gpurun_out/${TAG}_bench.err; cat gpurun_out/${TAG}_bench_1gpu.json
timeout 300 python tools/bench_configs.py > gpurun_out/${TAG}_configs.jsonl 2> gpurun_out/${TAG}_configs.err; cut -c1-330 gpurun_out/${TAG}_configs.jsonl
