#!/bin/bash
# instruction fetch counters of the front kernel (is the 60 KiB of its code an instruction-cache problem?)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_if; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $O/a -o p -- python $R/tools/exp_front.py --tag pmc --steps 1 --no-parity > $O/a.log 2>&1
cd $R; python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_if/a/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    if k.startswith("tk_k_front<") or k.startswith("tk_k_merge_all") or k.startswith("tk_k_place"):
        d = {c: max(v) for c, v in agg[k].items()}
        print(k[:30], {c: "%.4g" % v for c, v in sorted(d.items())}, "cycles per fetch %.1f" % (d.get("SQ_IFETCH_LEVEL", 0) / max(d.get("SQ_IFETCH", 1), 1)))
PY
