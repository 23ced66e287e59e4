#!/bin/bash
# Round 3, call N: SQ counters (separate --pmc passes, no tracing beside them) of the final kernels: the bench line's pipeline at 1 GiB and the
# generic engine's kernels at 256 MiB.
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out/pmc2 $R/gpurun_out/pmc_rx
cd /tmp && export TMPDIR=/tmp
date +%s > $R/gpurun_out/r3n_t0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"; do
  T=$(echo $SET | cut -d' ' -f1)
  timeout 100 rocprofv3 --pmc $SET --output-format csv -d $R/gpurun_out/pmc_rx/$T -o p -- python $R/bench.py --gpus 1 --steps 1 --warmup 0 --mib 256 --generic-engine --no-cpu-baseline --no-host-path > $R/gpurun_out/pmc_rx/$T.log 2>&1
  timeout 100 rocprofv3 --pmc $SET --output-format csv -d $R/gpurun_out/pmc2/$T -o p -- python $R/bench.py --gpus 1 --steps 1 --warmup 0 --mib 1024 --no-cpu-baseline --no-host-path > $R/gpurun_out/pmc2/$T.log 2>&1
done
cd $R; python - <<'PY'
import csv, glob, collections
for d in ("pmc_rx", "pmc2"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/{d}/*/p_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "tk_k_" in k:
                agg[k[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", d)
    for k, v in sorted(agg.items()):
        if max(v.get("SQ_INSTS_VALU", [0])) > 1e7:
            print("%-42s" % k, " ".join("%s=%.3g" % (c.replace("SQ_", ""), max(x)) for c, x in sorted(v.items())))
PY
find gpurun_out/pmc2 gpurun_out/pmc_rx -name '*.csv' -size +5M -delete
echo "elapsed $(( $(date +%s) - $(cat gpurun_out/r3n_t0) )) s"
