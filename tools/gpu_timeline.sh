#!/bin/bash
# usage: bash tools/gpu_timeline.sh CHUNK_MIB TAG -- kernel timeline (rocprofv3 --kernel-trace) of one profiled run, reduced to a per-kernel table
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tl_$2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
TIKTOKEN_AMD_CHUNK_BYTES=$(( $1 << 20 )) rocprofv3 --kernel-trace --output-format csv -d $O/trace -o tl -- python $R/tools/exp_front.py --tag tl --steps 2 --no-parity > $O/log.txt 2>&1
cd $R
f=$(find $O/trace -name '*kernel_trace.csv' | head -1)
python - "$f" "$O/timeline.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if r["Kernel_Name"].startswith(("tk_k", "void tk_k", "__amd_rocclr"))]
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
# the last full encode: from the last-but-N tk_k_mark_docs ... take the final 40% of the rows' time span
fronts = [r for r in rows if "tk_k_front<" in r["Kernel_Name"] and "false>" in r["Kernel_Name"].replace(" ", "")[-8:]]
out = open(sys.argv[2], "w")
t_end = rows[-1]["e"]
# find the start of the last step: the mark_docs launch that precedes the last group of fronts
marks = [r for r in rows if "tk_k_mark_docs" in r["Kernel_Name"]]
nchunks = max(1, len(marks) // 5)  # warmup + 2 timed + 2 profiled
t0 = marks[-nchunks]["s"]
print(f"chunks per step {nchunks}; last step spans {(t_end - t0) / 1e6:.3f} ms", file=out)
qk = "Queue_Id" if "Queue_Id" in rows[0] else None
for r in rows:
    if r["s"] < t0: continue
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]
    print(f'{(r["s"] - t0) / 1e3:10.1f} {(r["e"] - t0) / 1e3:10.1f} {(r["e"] - r["s"]) / 1e3:9.1f} us  q{r.get(qk, "?") if qk else "?":>3} {name}', file=out)
out.close()
PY
find $O/trace -name '*kernel_trace.csv' -size +8M -delete
head -5 $O/timeline.txt; wc -l $O/timeline.txt
