#!/bin/bash
# usage: bash tools/gpu_mid_trace.sh -- rocprofv3 kernel trace of tools/mid_call.py: the kernels of the LAST 4 KiB call with their start times
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/midtrace; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
MID_ONLY=4096 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o mt -- python $R/tools/mid_call.py > $O/log.txt 2>&1
cd $R
f=$(find $O/trace -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "tk_k" in r["Kernel_Name"] or "rocclr" in r["Kernel_Name"]]
for r in rows: r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
marks = [i for i, r in enumerate(rows) if "tk_k_mark_docs" in r["Kernel_Name"]]
i0 = marks[-3]  # a timed call (the last one is the profiled call)
t0 = rows[i0]["s"]
for r in rows[i0:marks[-2]]:
    print(f'{(r["s"] - t0) / 1e3:9.1f} {(r["e"] - r["s"]) / 1e3:8.1f} us  {r["Kernel_Name"].split("(")[0][:50]}  grid {r.get("Grid_Size")} wg {r.get("Workgroup_Size")}')
PY
find $O/trace -name '*.csv' -size +4M -delete
