#!/usr/bin/env python3
"""gpurun_out/pmc2/ (tools/gpu_pmc.sh: three rocprofv3 --pmc passes over bench.py --steps 1) -> profiles/<tag>_sq_counters.csv.
One row per kernel of the encode pipeline, value = the largest launch of that kernel."""
import collections
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from summarize_prof import canonical  # noqa: E402


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", "pmc2", "*", "p_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if k.startswith("void "):
                k = k[5:]
            k = canonical(k)
            if k.startswith("tk_k_"):
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    names = sorted({c for d in agg.values() for c in d})
    out = os.path.join(ROOT, "profiles", f"{tag}_sq_counters.csv")
    with open(out, "w") as fo:
        fo.write("# rocprofv3 --pmc (3 passes, tools/gpu_pmc.sh), bench.py --steps 1 --warmup 0 --mib 1024; value = largest launch of the kernel.\n")
        fo.write("# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are in quad-cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs.\n")
        fo.write("# VALU utilisation of a kernel = SQ_ACTIVE_INST_VALU / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 / 4).\n")
        fo.write("kernel," + ",".join(names) + "\n")
        for k in sorted(agg):
            fo.write(k + "," + ",".join("%.0f" % max(agg[k][c]) if c in agg[k] else "" for c in names) + "\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
