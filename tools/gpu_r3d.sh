#!/bin/bash
bash tools/gpu_ab.sh "merge_all||A=1" "per_bin||TIKTOKEN_AMD_DEBUG=8388608"
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
