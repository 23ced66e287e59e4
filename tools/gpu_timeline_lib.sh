#!/bin/bash
# usage: bash tools/gpu_timeline_lib.sh CHUNK_MIB TAG [VARIANT] -- tools/gpu_timeline.sh with a variant of the library (tools/build_variant.sh)
export TIKTOKEN_AMD_LIB=${3:+$GRAFT_REPO_ROOT/tiktoken_amd/csrc/variants/libtiktoken_amd_$3.so}
exec bash $GRAFT_REPO_ROOT/tools/gpu_timeline.sh "$1" "$2"
