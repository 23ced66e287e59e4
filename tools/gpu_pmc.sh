#!/bin/bash
# usage: bash tools/gpu_pmc.sh -- SQ counters (three passes) and cache counters (two passes: hits and misses of the L2, requests of the L1 to it) of
# bench.py --steps 1 --mib 1024, one rocprofv3 --pmc run per set; tools/summarize_pmc.py turns gpurun_out/pmc2 into profiles/<tag>_sq_counters.csv
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/pmc2
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z_0-9]+|GRBM_[A-Z_]+|TCP_[A-Za-z_0-9]+|TCC_[A-Za-z_0-9]+)\b" | sort -u > $R/gpurun_out/pmc2/counters.txt
wc -l $R/gpurun_out/pmc2/counters.txt
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum"; do
  T=$(echo $SET | cut -d' ' -f1)
  rocprofv3 --pmc $SET --output-format csv -d $R/gpurun_out/pmc2/$T -o p -- python $R/bench.py --gpus 1 --steps 1 --warmup 0 --mib 1024 --no-cpu-baseline --no-host-path > $R/gpurun_out/pmc2/$T.log 2>&1
done
cd $R; python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc2/*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("void "):
            k = k[5:]
        k = ("tk_k_front_slow" if k.split(">")[0].replace(" ", "").endswith(("true", ",1")) else "tk_k_front_given" if k.split(">")[0].replace(" ", "").endswith(",2") else "tk_k_front") if k.startswith("tk_k_front<") else k.split("<")[0]
        if k.startswith("tk_k_"):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-24s %16.0f" % (c, max(v)))
PY
find gpurun_out/pmc2 -name '*.csv' -size +5M -delete
