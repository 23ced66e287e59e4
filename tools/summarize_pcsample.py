#!/usr/bin/env python3
"""rocprofv3 PC-sampling CSVs -> small histograms (run on the GPU box by tools/gpu_pcsample.sh; the raw samples are too big to travel).

  hist_<method>_by_pc.csv       kernel, code-object offset, samples, share of the kernel's samples, instruction text [, issued / stall reason columns]
  hist_<method>_header.txt      the CSV's header and first rows (whatever this rocprofv3 writes)
"""
import collections
import csv
import glob
import os
import sys

src, dst, method = sys.argv[1], sys.argv[2], sys.argv[3]
files = [f for f in glob.glob(os.path.join(src, "**", "*.csv"), recursive=True)]
print("files:", [(os.path.basename(f), os.path.getsize(f)) for f in files])
kern_of = {}
for f in files:
    if f.endswith("kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            kern_of[r.get("Dispatch_Id")] = r.get("Kernel_Name", "?").split("(")[0][:60]
for f in files:
    if "pc_sampling" not in os.path.basename(f):
        continue
    with open(f) as fh:
        head = [next(fh, "") for _ in range(12)]
    open(os.path.join(dst, f"hist_{method}_header.txt"), "w").writelines(head)
    rd = csv.DictReader(open(f))
    cols = rd.fieldnames
    print("columns:", cols)
    inst_col = next((c for c in cols if c.lower() == "instruction"), None)
    off_col = next((c for c in cols if "offset" in c.lower()), None)
    disp_col = next((c for c in cols if c.lower() == "dispatch_id"), None)
    extra = [c for c in cols if any(k in c.lower() for k in ("issued", "reason", "type", "stall"))]
    hist = collections.Counter()
    text = {}
    ext = collections.defaultdict(collections.Counter)
    per_k = collections.Counter()
    n = 0
    for r in rd:
        n += 1
        k = kern_of.get(r.get(disp_col), "?") if disp_col else "?"
        key = (k, r.get(off_col) if off_col else r.get(inst_col))
        hist[key] += 1
        per_k[k] += 1
        if inst_col:
            text[key] = r.get(inst_col)
        for c in extra:
            ext[key][(c, r.get(c))] += 1
    print("samples:", n, "kernels:", per_k.most_common(12))
    with open(os.path.join(dst, f"hist_{method}_by_pc.csv"), "w") as fo:
        fo.write("kernel,offset,samples,share_of_kernel,instruction,detail\n")
        for (k, off), c in sorted(hist.items(), key=lambda kv: (kv[0][0], int(kv[0][1], 0) if kv[0][1] and kv[0][1].replace("0x", "").isalnum() and off_col else 0)):
            if per_k[k] < n * 0.002:
                continue
            det = ";".join(f"{a}={b}:{v}" for (a, b), v in ext[(k, off)].most_common(4))
            fo.write(f'"{k}",{off},{c},{c / per_k[k]:.5f},"{text.get((k, off), "")}","{det}"\n')
    print("wrote", os.path.join(dst, f"hist_{method}_by_pc.csv"), os.path.getsize(os.path.join(dst, f"hist_{method}_by_pc.csv")))
