#!/usr/bin/env python3
"""Turn a gpurun_out/prof_<tag>/ directory (tools/gpu_prof.sh) into the committed summaries under profiles/.

  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary, verbatim
  profiles/<tag>_hbm_traffic.csv    per kernel: launches, FETCH_SIZE / WRITE_SIZE per launch (KB, raw) and the
                                    corrected HBM bytes per launch
  profiles/traffic.json             what bench.py reads for `roofline.traffic`

Correction (MI355X_MICROARCH.md, HBM section): counters are in KB; on gfx950 FETCH_SIZE reports half of
the bytes actually fetched, so hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  The same run calibrates
it: tk_k_scan_small reads and rewrites the 1 MiB array of tile token counts and reports FETCH_SIZE = 515 KB and
WRITE_SIZE = 1024 KB (reads are counted at half, writes in full).  FETCH_SIZE and
WRITE_SIZE are collected in separate passes (they do not fit one pass: TCC has 4 slots).
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def canonical(name):
    """Kernel names as bench.py reports them: the pattern / special-token instances of the front kernel under one
    name, merge kernels by their piece-length class."""
    import re
    if name.startswith("tk_k_front<"):  # <pattern, specials, mode>: the two instances of the deferred tiles are reported under their own names
        args = name[len("tk_k_front<"):].split(">")[0].replace(" ", "").split(",")
        mode = args[2] if len(args) >= 3 else "0"
        return "tk_k_front_slow" if mode in ("true", "1") else ("tk_k_front_given" if mode == "2" else "tk_k_front")
    m = re.match(r"tk_k_merge_llane<(\d+)", name)
    if m:
        return "tk_k_merge_llane_" + m.group(1)
    m = re.match(r"tk_k_merge_group<(\d+)", name)
    if m:
        return "tk_k_merge_group_" + m.group(1)
    return name


def per_kernel(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0]
        if name.startswith("void "):
            name = name[5:]
        agg[canonical(name)].append(float(r["Counter_Value"]))
    return agg


def main():
    tag = sys.argv[1]
    mib = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    enc = sys.argv[3] if len(sys.argv) > 3 else "o200k_shaped"
    d = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    shutil.copy(os.path.join(d, "trace", f"{tag}_kernel_stats.csv"), os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
    f = per_kernel(os.path.join(d, "pmc_FETCH_SIZE", f"{tag}_counter_collection.csv"))
    w = per_kernel(os.path.join(d, "pmc_WRITE_SIZE", f"{tag}_counter_collection.csv"))
    out = {}
    with open(os.path.join(ROOT, "profiles", f"{tag}_hbm_traffic.csv"), "w") as fo:
        fo.write("kernel,launches,fetch_size_kb_per_launch_raw,write_size_kb_per_launch,hbm_bytes_per_launch_corrected\n")
        for k in sorted(set(f) | set(w)):
            if not k.startswith("tk_k_"):
                continue
            # per launch = the largest launch of that kernel (small launches of the same kernel, e.g. the
            # two tk_k_scan_small calls, are reported at their max)
            fk = max(f.get(k, [0.0]))
            wk = max(w.get(k, [0.0]))
            hbm = int((2 * fk + wk) * 1024)
            fo.write(f"{k},{len(f.get(k, []))},{fk:.1f},{wk:.1f},{hbm}\n")
            out[k] = {"hbm_bytes_per_launch": hbm, "fetch_kb_raw": fk, "write_kb": wk}
    import subprocess
    try:
        digest = open(os.path.join(d, "csrc_digest.txt")).read().strip()
    except OSError:
        digest = None
    git = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    dirty = bool(subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "tiktoken_amd/csrc"], capture_output=True, text=True).stdout.strip())
    json.dump({"tag": tag, "workload_mib": mib, "encoding": enc, "correction": "(2*FETCH_SIZE + WRITE_SIZE) * 1024",
               "csrc_digest": digest, "measured_at_git": git + ("+uncommitted changes under csrc/" if dirty else ""),
               "kernels": out}, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    print(open(os.path.join(ROOT, "profiles", f"{tag}_hbm_traffic.csv")).read())


if __name__ == "__main__":
    main()
