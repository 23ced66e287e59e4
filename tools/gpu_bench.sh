#!/bin/bash
# usage: bash tools/gpu_bench.sh [MIB] [STEPS]  -- smoke + bench + rocprofv3 kernel stats (run on the GPU box via gpurun)
MIB=${1:-1024}; STEPS=${2:-3}
mkdir -p gpurun_out
export TMPDIR=/tmp
python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
tail -3 gpurun_out/smoke.log
python bench.py --gpus 1 --steps $STEPS --warmup 1 --mib $MIB > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"
tail -c 3000 gpurun_out/bench.json; tail -5 gpurun_out/bench.err
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 2 --warmup 1 --mib $MIB --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1
cd $GRAFT_REPO_ROOT; ls -R gpurun_out/prof | head -20
f=$(find gpurun_out/prof -name '*kernel_stats.csv' | head -1); echo "stats file: $f"; head -20 "$f"
