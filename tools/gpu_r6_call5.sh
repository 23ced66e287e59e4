#!/bin/bash
# round 6, call 5: the host not waiting for the deferred tiles' counters (default) against waiting ($TIKTOKEN_AMD_DEFER_SYNC=1), one box
python -m pytest tests/test_gpu_parity.py -x -q -k "give_up or gives_up or certain_start or long or baseline" 2>&1 | tail -4
bash tools/gpu_ab.sh "nowait||" "wait||TIKTOKEN_AMD_DEFER_SYNC=1"
for m in "" 1; do echo "DEFER_SYNC=$m"; TIKTOKEN_AMD_DEFER_SYNC=$m python tools/bench_configs.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: j = json.loads(l)
    except Exception: continue
    print(j['config'][:24], j['ms_per_step'], j.get('GBps'), j.get('parity'))
"; done
