#!/bin/bash
# usage: bash tools/gpu_pmc_kernel.sh KERNEL_SUBSTRING TAG [env...] -- SQ / cache counters of one kernel (separate --pmc passes), 1 GiB bench corpus
K=$1; TAG=$2; shift 2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmck_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_ATOMIC_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  env "$@" timeout 200 rocprofv3 --pmc $SET --output-format csv -d $O/p$i -o p -- python $R/tools/exp_front.py --tag pmc --steps 1 --no-parity > $O/p$i.log 2>&1
done
cd $R; python - "$O" "$K" <<'PY'
import csv, glob, sys, collections
O, K = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{O}/p*/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if K in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-36s n=%d max %16.0f  sum/launchgroup %16.0f" % (c, len(v), max(v), sum(v) / max(1, len(v)) ))
PY
find $O -name '*.csv' -size +5M -delete
