#!/bin/bash
# T2 (tk_encode_batch: pageable host text in, ids in host memory out) under the knobs of the host path, 1 GiB, five runs each (the last two shown)
for cfg in "" "TIKTOKEN_AMD_H2D_BLOCK_MIB=32" "TIKTOKEN_AMD_H2D_BLOCK_MIB=128" "TIKTOKEN_AMD_H2D_BLOCK_MIB=128 TIKTOKEN_AMD_HOST_CHUNK_MIB=64" "TIKTOKEN_AMD_H2D_BLOCK_MIB=256 TIKTOKEN_AMD_HOST_CHUNK_MIB=32"; do
  echo "== $cfg"; env $cfg timeout 200 python tools/host_path_check.py 1024 2>&1 | grep "^run" | cut -c1-80 | tail -2
done
