#!/bin/bash
# usage: tools/grun.sh TIMEOUT 'command'  -- rebuild every native piece, then run the command on the GPU box (never ship a stale .so)
set -e
cd "$(dirname "$0")/.."
make -s -C tiktoken_amd/csrc libtiktoken_amd.so libtkcorpus.so
make -s -C oracle libtk_oracle.so
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
