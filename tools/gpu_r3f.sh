#!/bin/bash
# round 3, call F: small calls from several threads, the new pattern features, segment size of the generic engine's speculative pass
O=gpurun_out/r3f; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_api.py tests/test_gpu_regex.py -m gpu -x -q --timeout 150 2>&1 | tail -6 > $O/pytest.log; cat $O/pytest.log
timeout 200 python tools/small_call.py > $O/small_calls.txt 2>&1; cat $O/small_calls.txt
for k in 8 9 10 11; do
  TIKTOKEN_AMD_RX_SEG_SHIFT=$k timeout 150 python bench.py --generic-engine --mib 256 --steps 3 --warmup 1 --no-host-path --no-cpu-baseline > $O/gen_$k.json 2> $O/gen_$k.err
  python - $k <<'PY'
import json,sys
k=sys.argv[1]
try:
    j=json.loads(open(f'gpurun_out/r3f/gen_{k}.json').read().strip().splitlines()[-1]); km=j["roofline"]["kernels_ms_avg"]
    print("seg_shift",k,"GB/s",j["value"],"speculate",km.get("tk_k_rx_speculate"),"resolve",km.get("tk_k_rx_resolve"))
except Exception as e: print("seg_shift",k,"failed",e)
PY
done
