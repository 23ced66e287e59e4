#!/bin/bash
# usage (on the GPU box, through tools/grun.sh; build the variant FIRST, here: tools/build_variant.sh small1 -DTK_SMALL_ONE_PHASE=1):
#   bash tools/gpu_small_variant.sh
# The experiment prepared at the end of round 4 and never run: tk_k_small with the one-lane and the sixteen-lane merges in ONE phase.
# 1. parity of the variant (the small / mid-size GPU tests + tools/mid_corpus.py's 1578 documents), 2. web-text timings: shipped library,
# variant, variant with the mid-size segments keeping their long pieces (debug bit 0x20000000).  Decide by profiles/r04_mid_calls_corpus.txt's
# table: the variant has to beat 250 us at 2 KiB (the pipeline: 160) and the pipeline at 16 .. 128 KiB before it becomes the shipped form.
V=$PWD/tiktoken_amd/csrc/variants/libtiktoken_amd_small1.so
[ -f "$V" ] || { echo "build the variant first: tools/build_variant.sh small1 -DTK_SMALL_ONE_PHASE=1"; exit 1; }
mkdir -p gpurun_out
TIKTOKEN_AMD_LIB=$V timeout 200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -q -x -k "small or mid_size or hello_world or share_the_slots" 2>&1 | tail -3
for spec in "shipped||" "one_phase|$V|" "one_phase_mid_keeps_long|$V|TIKTOKEN_AMD_DEBUG=536870912"; do
  IFS='|' read -r tag lib envs <<< "$spec"
  echo "== $tag"
  env $envs ${lib:+TIKTOKEN_AMD_LIB=$lib} timeout 120 python tools/mid_corpus.py 2>&1 | tee gpurun_out/small_variant_$tag.txt | tail -10
done
