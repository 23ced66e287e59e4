#!/bin/bash
# usage: bash tools/gpu_prof.sh [MIB] [TAG] -- rocprofv3 kernel-trace stats + HBM traffic counters (separate passes)
MIB=${1:-1024}; TAG=${2:-r01}
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/prof_$TAG
python $R/bench.py --csrc-digest > $R/gpurun_out/prof_$TAG/csrc_digest.txt  # (the sources these counters are measured on: bench.py ties roofline.traffic to it)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG/trace -o $TAG -- python $R/bench.py --gpus 1 --steps 3 --warmup 1 --mib $MIB --no-cpu-baseline --no-host-path > $R/gpurun_out/prof_$TAG/trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/prof_$TAG/pmc_$C -o $TAG -- python $R/bench.py --gpus 1 --steps 1 --warmup 0 --mib $MIB --no-cpu-baseline --no-host-path > $R/gpurun_out/prof_$TAG/pmc_$C.log 2>&1
done
cd $R; find gpurun_out/prof_$TAG -type f | head -30
f=$(find gpurun_out/prof_$TAG/trace -name '*kernel_stats.csv' | head -1); echo "== $f"; cat "$f" | head -20
f=$(find gpurun_out/prof_$TAG/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1); echo "== $f"; head -5 "$f"
# keep only small files for the merge-back
find gpurun_out/prof_$TAG -name '*kernel_trace.csv' -size +20M -delete
