#!/bin/bash
# usage: bash tools/gpu_pmc_ab.sh "tag|variant|ENV=.." ...  -- SQ instruction counters of the front kernel per library variant (one rocprofv3 --pmc pass each, exp_front.py)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_ab; mkdir -p $O; V=$R/tiktoken_amd/csrc/variants
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  IFS='|' read -r tag lib envs <<< "$spec"
  env $envs TIKTOKEN_AMD_LIB=${lib:+$V/libtiktoken_amd_$lib.so} timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/$tag -o p -- python $R/tools/exp_front.py --tag $tag --steps 1 --no-parity > $O/$tag.log 2>&1
done
cd $R; python - "$O" "$@" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
for spec in sys.argv[2:]:
    tag = spec.split("|")[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{O}/{tag}/**/p_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(agg):
        if not (k.startswith("tk_k_front<") or k.startswith("tk_k_merge_all") or k.startswith("tk_k_place")):
            continue
        d = agg[k]
        print("%-8s %-28s " % (tag, k[:28]) + " ".join("%s=%.4g" % (c.replace("SQ_", ""), max(v)) for c, v in sorted(d.items())))
PY
find $O -name '*.csv' -size +5M -delete
