#!/bin/bash
# usage (on the GPU box, through tools/grun.sh; build the variants FIRST, here:
#   tools/build_variant.sh small1 -DTK_SMALL_ONE_PHASE=1; tools/build_variant.sh small2 -DTK_SMALL_ONE_PHASE=2 [-DTK_SMALL_K=8]):
#   bash tools/gpu_small_variant.sh
# The experiment prepared at the end of round 4 and never run: tk_k_small with the one-lane and the sixteen-lane merges in ONE phase.
# 1. parity of the variant (the small / mid-size GPU tests + tools/mid_corpus.py's 1578 documents), 2. web-text timings: shipped library,
# variant, variant with the mid-size segments keeping their long pieces (debug bit 0x20000000).  Decide by profiles/r04_mid_calls_corpus.txt's
# table: the variant has to beat 250 us at 2 KiB (the pipeline: 160) and the pipeline at 16 .. 128 KiB before it becomes the shipped form.
# Level 2 (small2): ... and TK_SMALL_K merges per round of probes in the sixteen-lane merge (profiles/r04_merge_steps_sim.txt: 2.9 per round for K = 4).
V=$PWD/tiktoken_amd/csrc/variants/libtiktoken_amd_small1.so
V2=$PWD/tiktoken_amd/csrc/variants/libtiktoken_amd_small2.so
[ -f "$V" ] && [ -f "$V2" ] || { echo "build the variants first: tools/build_variant.sh small1 -DTK_SMALL_ONE_PHASE=1; tools/build_variant.sh small2 -DTK_SMALL_ONE_PHASE=2"; exit 1; }
mkdir -p gpurun_out
for lib in $V $V2; do
  TIKTOKEN_AMD_LIB=$lib timeout 200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -q -x -k "small or mid_size or hello_world or share_the_slots" 2>&1 | tail -3
done
for spec in "shipped||" "one_phase|$V|" "one_phase_mid_keeps_long|$V|TIKTOKEN_AMD_DEBUG=536870912" "k_merges|$V2|" "k_merges_mid_keeps_long|$V2|TIKTOKEN_AMD_DEBUG=536870912"; do
  IFS='|' read -r tag lib envs <<< "$spec"
  echo "== $tag"
  env $envs ${lib:+TIKTOKEN_AMD_LIB=$lib} timeout 120 python tools/mid_corpus.py 2>&1 | tee gpurun_out/small_variant_$tag.txt | tail -10
done
