#!/bin/bash
GPU_MAX_HW_QUEUES=4 bash tools/gpu_timeline.sh 1024 c1024q4 > /dev/null 2>&1
GPU_MAX_HW_QUEUES=16 bash tools/gpu_timeline.sh 1024 c1024q16 > /dev/null 2>&1
for t in c1024q4 c1024q16; do echo "== $t"; head -1 gpurun_out/tl_$t/timeline.txt; grep -v "fillBuffer\|copyBuffer" gpurun_out/tl_$t/timeline.txt | sed -n 2,60p; done
