#!/bin/bash
# Round 5, GPU call 3: tk_k_rx_speculate_staged with 256 (shipped) / 128 / 64 segments per workgroup, then its SQ counters.
TAG=r05
mkdir -p gpurun_out
V=$PWD/tiktoken_amd/csrc/variants
O=gpurun_out/${TAG}_generic_staged_wg.txt; : > $O
run() {  # tag, lib, staged
  TIKTOKEN_AMD_LIB=${2:+$V/libtiktoken_amd_$2.so} TIKTOKEN_AMD_RX_STAGED=$3 timeout 200 python bench.py --generic-engine --mib 256 --steps 3 --warmup 1 --no-host-path --no-hf --cpu-sample-mib 32 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read()); k = j['roofline']['kernels_ms_avg']
print('$1 256 MiB: %.2f GB/s  %.3f ms  speculate %.3f link %.3f resolve %.3f merge %.3f front %.3f  parity %s' % (j['value'], j['ms_per_step'], k.get('tk_k_rx_speculate', 0), k.get('tk_k_rx_link', 0), k.get('tk_k_rx_resolve', 0), k.get('tk_k_rx_merge', 0), k.get('tk_k_front', 0), j['parity_all_tokens_vs_oracle']))
" >> $O
}
for rep in 1 2; do
  run one-loop "" 0; run wg256 "" 1; run wg128 s128 1; run wg64 s64 1
done
cat $O
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/pmc_rx
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $SET --output-format csv -d $R/gpurun_out/pmc_rx/p$i -o p -- python $R/bench.py --generic-engine --mib 256 --steps 1 --warmup 0 --no-host-path --no-hf --no-cpu-baseline > $R/gpurun_out/pmc_rx/p$i.log 2>&1
done
cd $R; python - <<'PY' | tee -a gpurun_out/r05_generic_staged_wg.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_rx/*/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("tk_k_rx"):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-24s %16.0f" % (c, max(v)))
PY
find gpurun_out/pmc_rx -name '*.csv' -size +2M -delete
