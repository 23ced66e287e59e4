#!/bin/bash
# usage (GPU box): bash tools/gpu_pcsample.sh [METHOD] [INTERVAL] [MIB] -- PC sampling of the encode pipeline (rocprofv3 --pc-sampling-beta-enabled) over
# tools/exp_front.py; the samples are aggregated on the box (tools/summarize_pcsample.py) into gpurun_out/pcs/hist_*.csv: samples per code-object offset,
# per kernel, with the instruction text -- what an instruction diet of the front kernel is steered by.  Wrapped in a timeout of its own.
METHOD=${1:-stochastic}; INTERVAL=${2:-1048576}; MIB=${3:-1024}
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/pcs
cd /tmp && export TMPDIR=/tmp
UNIT=cycles; [ "$METHOD" = host_trap ] && UNIT=time
timeout 420 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $UNIT --pc-sampling-method $METHOD --pc-sampling-interval $INTERVAL --kernel-trace \
  --output-format csv -d /tmp/pcs_out -o pcs -- python $R/tools/exp_front.py --tag pcs --mib $MIB --steps 4 --no-parity > $R/gpurun_out/pcs/run_$METHOD.log 2>&1
echo "rocprofv3 rc=$?"; tail -3 $R/gpurun_out/pcs/run_$METHOD.log | cut -c1-300
find /tmp/pcs_out -type f | head; du -sh /tmp/pcs_out
cd $R; python tools/summarize_pcsample.py /tmp/pcs_out gpurun_out/pcs $METHOD
