#!/bin/bash
# registers / scratch / LDS of every kernel (hipcc remarks), to be looked at BEFORE a change goes to the GPU: scratch > 16 bytes on
# tk_k_front means a second inlined copy of the scanner made the compiler spill.
cd "$(dirname "$0")/../tiktoken_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage -c tk_api.hip -o /tmp/tk_api_chk.o 2>&1 |
  sed 's/ \[-Rpass-analysis=kernel-resource-usage\]//' |
  awk '/Function Name/ {name=$NF} / VGPRs:/ {v=$NF} /ScratchSize/ {s=$NF} /VGPRs Spill/ {sp=$NF} /LDS Size/ {printf "%-60s vgpr %3s scratch %4s spill %2s lds %s\n", substr(name,1,60), v, s, sp, $NF}'
