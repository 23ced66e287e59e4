#!/bin/bash
# usage (GPU box): bash tools/gpu_bench_n2_dry.sh [MIB]  -- the N > 1 leg of bench.py executed on a ONE-GPU box: two ranks share device 0, the
# collectives and the gather of the ids go through host copies over gloo ($TIKTOKEN_AMD_BENCH_BACKEND).  What it proves: the leg runs end
# to end, every rank's shard is compared with the oracle, rank 0 verifies what it received (gather_verified).  What it does NOT: RCCL, xGMI,
# any rate (the value printed is meaningless: two encoders on one device, ids through the host).
MIB=${1:-256}
cd "$(dirname "$0")/.."
TIKTOKEN_AMD_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 2 --warmup 1 --mib $MIB
echo "torchrun rc=$?"
