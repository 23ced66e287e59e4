#!/bin/bash
# round 3, call G: the generic engine with the link pass and the wavefront-per-document resolve; small calls from several threads
O=gpurun_out/r3g; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_regex.py tests/test_gpu_api.py -m gpu -x -q --timeout 150 2>&1 | tail -6 > $O/pytest.log; cat $O/pytest.log
timeout 100 python tools/rx_diag.py > $O/rx_diag.txt 2>&1; cut -c1-330 $O/rx_diag.txt
TIKTOKEN_AMD_DEBUG=$((0x4000000)) timeout 100 python tools/rx_diag.py '\w+|[^\w\s]+|\s+' fuzz > $O/rx_diag_lane_per_doc.txt 2>&1; cut -c1-330 $O/rx_diag_lane_per_doc.txt | tail -3
for k in 6 7 8; do
  TIKTOKEN_AMD_RX_SEG_SHIFT=$k timeout 150 python bench.py --generic-engine --mib 256 --steps 3 --warmup 1 --no-host-path --no-hf > $O/gen_$k.json 2> $O/gen_$k.err
  python - $k <<'PY'
import json,sys
k=sys.argv[1]
try:
    j=json.loads(open(f'gpurun_out/r3g/gen_{k}.json').read().strip().splitlines()[-1]); km=j["roofline"]["kernels_ms_avg"]
    print("seg_shift",k,"GB/s",j["value"],"parity",j["parity_all_tokens_vs_oracle"],"speculate",km.get("tk_k_rx_speculate"),"link",km.get("tk_k_rx_link"),"resolve",km.get("tk_k_rx_resolve"))
except Exception as e: print("seg_shift",k,"failed",e)
PY
done
