#!/usr/bin/env python3
"""Kernel experiments on the GPU box: one configuration of the library (selected through $TIKTOKEN_AMD_LIB / $TIKTOKEN_AMD_FRONT_WGS /
$TIKTOKEN_AMD_DEBUG) on the bench corpus; prints one JSON line with wall time per step, per-kernel HIP-event times and whether EVERY token equals the oracle's (computed once per box and kept in /tmp).

    python tools/exp_front.py --tag NAME [--mib 1024] [--steps 3]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="default")
    ap.add_argument("--mib", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--encoding", default="o200k_shaped")
    ap.add_argument("--no-parity", action="store_true")
    args = ap.parse_args()
    import torch
    from bench import gen_corpus, KERNELS, DevArray
    from tiktoken_amd._tiktoken import CoreBPE
    from tiktoken_ext import amd_shaped

    nbytes = args.mib << 20
    ncpu = len(os.sched_getaffinity(0))
    blob, doc_off = gen_corpus(0x5EED0003, 1, nbytes, min(ncpu, 32))
    n_docs = len(doc_off) - 1
    spec = amd_shaped.ENCODING_CONSTRUCTORS[args.encoding]()
    core = CoreBPE(spec["mergeable_ranks"], spec["special_tokens"], spec["pat_str"], device=0)
    d_text = torch.from_numpy(blob).cuda()
    d_off = torch.from_numpy(doc_off.view(np.int64)).cuda()
    torch.cuda.synchronize()
    run = lambda: core.encode_batch_device(d_text.data_ptr(), nbytes, d_off.data_ptr(), doc_off, n_docs)
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        dt, nt, do = run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    host = {k: core.stat(k) for k in ("chunks", "back_streams", "host_front_us", "host_back_us", "host_back_wait_us", "host_finish_us", "host_tail_us", "host_total_us")}
    core.set_profiling(True)
    core.reset_kernel_ms()
    for _ in range(2):
        run()
    core.set_profiling(False)
    kern = {}
    for k in KERNELS:
        kms, n = core.kernel_ms(k)
        if n:
            kern[k] = round(kms / n, 4)
    out = {"tag": args.tag, "mib": args.mib, "ms_per_step": round(ms, 3), "gbps": round(nbytes / ms / 1e6, 2), "kernels_ms": kern,
           "kernels_sum_ms": round(sum(kern.values()), 3), "tokens": int(nt),
           "front_wgs": core.stat("front_wgs_per_cu"), "pieces": core.last_stats()["pieces"],
           "host": host, "lib": os.environ.get("TIKTOKEN_AMD_LIB", ""), "dbg": os.environ.get("TIKTOKEN_AMD_DEBUG", "")}
    if core.stat("time_15") or os.environ.get("TKF_TIMING"):  # (a -DTKF_TIMING build: cycles per tile and phase, thread 0's clock at the phase boundaries)
        core.stat("time_reset")
        run()
        torch.cuda.synchronize()
        tiles = max(core.stat("time_15"), 1)
        out["cycles_per_tile"] = {str(i): round(core.stat(f"time_{i}") / tiles, 1) for i in range(14)}
        ts = max(core.stat("time_s15"), 1)
        out["deferred_tiles"], out["cycles_per_deferred_tile"] = ts, {str(i): round(core.stat(f"time_s{i}") / ts, 1) for i in range(14)}
        d = core.stat("time_14")
        out["tiles"], out["deferred_no_certain_start_in_left_context"], out["deferred_piece_leaves_window"] = tiles, d & 0xFFFFFFFF, d >> 32
    if not args.no_parity:
        cache = f"/tmp/tk_oracle_{args.encoding}_{args.mib}.npz"
        if os.path.exists(cache):
            z = np.load(cache)
            ctoks, coff = z["t"], z["o"]
        else:
            from oracle import c_oracle
            pat_id = {"gpt2_shaped": 0, "cl100k_shaped": 1, "o200k_shaped": 2}[args.encoding]
            C = c_oracle.COracle(pat_id, spec["mergeable_ranks"], spec["special_tokens"])
            ctoks, coff = C.encode_batch(blob[:nbytes], doc_off, None, ncpu)
            ctoks, coff = np.array(ctoks), np.array(coff)
            np.savez(cache, t=ctoks, o=coff)
        g_off = torch.as_tensor(DevArray(do, n_docs + 1, "<i8"), device="cuda").cpu().numpy().astype(np.uint64)
        g_tok = torch.as_tensor(DevArray(dt, max(nt, 1), "<i4"), device="cuda")[:nt].cpu().numpy().view(np.uint32)
        out["parity"] = bool(np.array_equal(g_off, coff) and np.array_equal(g_tok, ctoks))
    print("EXP " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
