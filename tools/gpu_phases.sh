#!/bin/bash
# usage: bash tools/gpu_phases.sh TAG -- per-phase cost of tk_k_front: the kernel is stopped after each phase by a debug bit
# (TIKTOKEN_AMD_DEBUG 0x1000 .. 0x10000: nothing behind the front kernel runs then, tk_api.hip stage_back `front_only`; 2: every probe
# answered without a table access; 8: pieces that are not tokens are not claimed), timed with the library's HIP events; SQ_INSTS_* counted
# in one rocprofv3 --pmc pass per variant.  The series stops at the first variant that fails (a faulting kernel can leave the box unusable).
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/phases_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
VARS=${VARIANTS:-0 0x1000 0x2000 0x4000 0x8000 0x10000 2 8}
for V in $VARS; do
  D=$((V))
  TIKTOKEN_AMD_DEBUG=$D timeout 90 python $R/bench.py --gpus 1 --steps 2 --warmup 1 --mib 1024 --no-cpu-baseline --no-host-path > $O/bench_$V.json 2> $O/bench_$V.err || { echo "variant $V failed: $(tail -2 $O/bench_$V.err)"; break; }
done
for V in ${PMC_VARIANTS:-$VARS}; do
  D=$((V))
  [ -s $O/bench_$V.json ] || continue
  TIKTOKEN_AMD_DEBUG=$D timeout 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_$V -o p -- python $R/bench.py --gpus 1 --steps 1 --warmup 0 --mib 1024 --no-cpu-baseline --no-host-path > $O/pmc_$V.log 2>&1 || { echo "pmc pass $V failed"; break; }
done
cd $R; python - "$O" $VARS <<'PY'
import csv, glob, json, os, sys, collections
O = sys.argv[1]
print("variant,front_ms,all_kernels_ms,VALU,SALU,LDS,VMEM_RD,WAVE_CYCLES,WAIT_ANY,WAIT_INST_ANY,ACTIVE_VALU")
for V in sys.argv[2:]:
    try:
        j = json.loads(open(f"{O}/bench_{V}.json").read().strip().splitlines()[-1])
        fm = j["roofline"]["kernels_ms_avg"].get("tk_k_front"); am = j["roofline"]["all_kernels_ms_per_step"]
    except Exception as e:
        fm = am = None
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{O}/pmc_{V}/**/p_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "tk_k_front<" in r["Kernel_Name"] and r["Kernel_Name"].split("(")[0].replace(" ", "").endswith(",0>"):
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    g = lambda c: ("%.0f" % max(agg[c])) if agg.get(c) else ""
    print(",".join(map(str, [V, fm, am, g("SQ_INSTS_VALU"), g("SQ_INSTS_SALU"), g("SQ_INSTS_LDS"), g("SQ_INSTS_VMEM_RD"), g("SQ_WAVE_CYCLES"), g("SQ_WAIT_ANY"), g("SQ_WAIT_INST_ANY"), g("SQ_ACTIVE_INST_VALU")])))
PY
find $O -name '*.csv' -size +5M -delete
