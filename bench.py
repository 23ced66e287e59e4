#!/usr/bin/env python3
"""Headline benchmark: GB/s of UTF-8 text encoded by the MI355X BPE path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is one pass of the hot path (encode_ordinary_batch semantics, reference
tiktoken/core.py:164-176 / src/lib.rs:360-373) over one batch of synthetic documents that is
already resident in HBM: the o200k-shaped encoding on a 1 GiB synthetic web-text corpus per GPU
(BASELINE.json configs[2]; seeds and mix per SURVEY.md 8(d)).  With N > 1 every rank encodes its
own 1 GiB shard (weak scaling, documents never interact) and rank 0 gathers the token-id buffers
with one padded RCCL gather per step; the gather of step k overlaps with the encode of step k + 1 and
the last one is waited for inside the timed region.

Protocol = the reference's scripts/benchmark.py:15-26: bytes = sum of UTF-8 lengths, warm-up first,
wall clock around the timed calls (here: barrier + synchronize on both sides, max over ranks).
Rank 0 prints ONE JSON line; `roofline` is for the dominant kernel (HIP-event durations measured
here, on the stream the kernels run on), `cpu_baseline` is the C oracle timed on the host cores
over a bounded sample of the same corpus; the WHOLE GPU result (every document, every token of the
1 GiB) is compared with the oracle.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md)
KERNELS = ["tk_k_mark_docs", "tk_k_front", "tk_k_front_slow", "tk_k_front_given", "tk_k_single_front", "tk_k_bincount", "tk_k_binfill", *[f"tk_k_merge_llane_{i}" for i in (16, 24, 32, 48, 64)],
           *[f"tk_k_merge_group_{i}" for i in (8, 16, 32, 64)], "tk_k_merge_all", "tk_k_merge_rounds", "tk_k_merge_rounds_wide", "tk_k_merge_long", "tk_k_count_tiles", "tk_k_scan_small", "tk_k_scan_sums", "tk_k_scan_apply", "tk_k_place", "tk_k_docoff",
           "tk_k_rx_speculate", "tk_k_rx_link", "tk_k_rx_resolve", "tk_k_rx_merge"]  # (the last four only run for a pat_str on the generic engine)


def gen_corpus(seed: int, mix: int, nbytes: int, threads: int):
    lib = ctypes.CDLL(os.path.join(ROOT, "tiktoken_amd", "csrc", "libtkcorpus.so"))
    lib.tkc_generate.argtypes = [ctypes.c_uint64, ctypes.c_int, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p,
                                 ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int]
    out = np.zeros(nbytes + 64, np.uint8)
    maxd = nbytes // 64 + 2
    off = np.empty(maxd + 1, np.uint64)
    nd = ctypes.c_uint64()
    rc = lib.tkc_generate(seed, mix, nbytes, out.ctypes.data, off.ctypes.data, maxd, ctypes.byref(nd), threads)
    assert rc == 0, rc
    return out, off[: nd.value + 1].copy()


def hf_tokenizers_rate(encoding, spec, blob, doc_off, nbytes, ctoks, coff, sample_mib=32):
    """Context for `cpu_baseline`: the reference's Rust core cannot be built here, so beside the C restatement stands a
    Rust tokenizer that can run -- HF `tokenizers`, built from the same vocabulary by transformers' TikTokenConverter (the merge list
    reconstructed from the ranks), `encode_batch` over all host threads on the first `sample_mib` MiB.  Its ids must be the oracle's.
    Never the baseline itself; None with the reason when the packages are missing."""
    try:
        import base64
        import tempfile
        import types

        def load_tiktoken_bpe(path, expected_hash=None):  # (the converter reads the vocabulary through `tiktoken.load`)
            return {base64.b64decode(t): int(r) for t, r in (line.split() for line in open(path, "rb").read().splitlines() if line)}

        if "tiktoken" not in sys.modules:
            shim, shim_load = types.ModuleType("tiktoken"), types.ModuleType("tiktoken.load")
            shim_load.load_tiktoken_bpe = load_tiktoken_bpe
            shim.load = shim_load
            sys.modules["tiktoken"], sys.modules["tiktoken.load"] = shim, shim_load
        import tokenizers
        from transformers.convert_slow_tokenizer import TikTokenConverter

        pat = spec["pat_str"]
        if "{1,3}+" in pat:  # Oniguruma reads {1,3}+ as a repeated repeat and $ as end of line: the same pattern in its dialect
            pat = (pat.replace("?+", "?").replace("++", "+").replace("*+", "*").replace("{1,3}+", "{1,3}").replace("$", "\\z"))
        with tempfile.NamedTemporaryFile("wb", suffix=".tiktoken", delete=False) as f:
            for tok, rank in sorted(spec["mergeable_ranks"].items(), key=lambda kv: kv[1]):
                f.write(base64.b64encode(tok) + b" " + str(rank).encode() + b"\n")
            path = f.name
        hf = TikTokenConverter(vocab_file=path, pattern=pat, add_prefix_space=False).converted()
        os.unlink(path)
        nd = max(int(np.searchsorted(doc_off, min(nbytes, sample_mib << 20), side="right")) - 1, 1)
        sb = int(doc_off[nd])
        raw = blob[:sb].tobytes()
        texts = [raw[int(doc_off[d]):int(doc_off[d + 1])].decode() for d in range(nd)]
        t0 = time.perf_counter()
        enc = hf.encode_batch(texts, add_special_tokens=False)
        dt = time.perf_counter() - t0
        same = all(e.ids == ctoks[int(coff[d]):int(coff[d + 1])].tolist() for d, e in list(enumerate(enc))[:: max(1, nd // 2000)])
        return {"value": round(sb / dt / 1e9, 4), "unit": "GB/s", "what": f"HF tokenizers {tokenizers.__version__} (Rust BPE + Oniguruma) encode_batch, "
                f"first {nd} documents ({sb} bytes), all host threads, one run", "same_ids_as_oracle_on_a_sample_of_documents": bool(same)}
    except Exception as e:  # (context only: never fails the bench)
        return {"value": None, "why": f"{type(e).__name__}: {e}"[:200]}


def csrc_digest() -> str:
    """sha256 over the product's native sources (tiktoken_amd/csrc: *.hip *.h *.cpp *.inc Makefile, names and contents): what `roofline.traffic`
    -- read from profiles/traffic.json, not measured in this run -- is tied to.  tools/gpu_prof.sh records it beside the counters."""
    import hashlib

    d = os.path.join(ROOT, "tiktoken_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".cpp", ".inc")) or name == "Makefile":
            h.update(name.encode() + b"\0")
            with open(os.path.join(d, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


def host_cores(ncpu_affinity: int) -> float:
    """CPU cores this process may actually use: min(scheduler affinity, cgroup v2 quota).  A box that shows 64 CPUs to a container whose
    cpu.max is "1600000 100000" gives it 16 cores' worth of time however many threads run."""
    q = _read_first("/sys/fs/cgroup/cpu.max")
    try:
        quota, period = q.split()
        if quota != "max":
            return min(float(ncpu_affinity), round(int(quota) / int(period), 2))
    except Exception:
        pass
    return float(ncpu_affinity)


def link_rates():
    """What the host-device link gives on THIS box (tools/ubench/pcie_rates.hip: page-locked H2D, D2H, both at once, 1 GiB in blocks of 128 MiB; the
    staging alternatives beside them): the ceiling of the host-buffer entry T2.  Built on first use when the binary is not there; None with
    the reason when that fails (never fails the bench)."""
    import subprocess

    exe = os.path.join(ROOT, "tools", "ubench", "pcie_rates")
    try:
        if not os.path.exists(exe):
            subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "--offload-arch=gfx950", exe + ".hip", "-pthread", "-o", exe], stderr=subprocess.DEVNULL, timeout=300)
        r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    except Exception as e:
        return {"error": f"{type(e).__name__}: {str(e)[:160]}"}


def _read_first(path):
    try:
        return open(path).readline().strip()
    except OSError:
        return None


class DevArray:
    """Zero-copy view of library-owned device memory for torch (via __cuda_array_interface__)."""

    def __init__(self, ptr: int, n: int, typestr: str):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def one_process(args, torch):
    """bench.py --gpus N --one-process: the product's own several-GPU path.  N document shards of --mib MiB each, host text in (the group's
    contract), every shard encoded on its device, ids gathered on the first device; value = PCIe-inclusive aggregate rate."""
    import tiktoken_amd  # noqa: F401
    from tiktoken_amd._tiktoken import CoreBPE
    from tiktoken_ext import amd_shaped

    n = args.gpus
    have = torch.cuda.device_count()
    devices = list(range(n)) if have >= n else [i % max(have, 1) for i in range(n)]
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    parts, offs, base = [], [np.zeros(1, np.uint64)], 0
    for r in range(n):
        b, o = gen_corpus(0x5EED0004 + r, 1, args.mib << 20, min(ncpu, 32))
        parts.append(b[: int(o[-1])])
        offs.append(o[1:] + np.uint64(base))
        base += int(o[-1])
    blob = np.concatenate(parts + [np.zeros(64, np.uint8)])[:base]
    doc_off = np.concatenate(offs)
    spec = amd_shaped.ENCODING_CONSTRUCTORS[args.encoding]()
    core = CoreBPE(spec["mergeable_ranks"], spec["special_tokens"], spec["pat_str"], devices=devices)
    for _ in range(args.warmup):
        core.encode_batch_gathered(blob, doc_off)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        dt, nt, do = core.encode_batch_gathered(blob, doc_off)
    el = time.perf_counter() - t0
    # what device 0 holds after the gather against the oracle's encoding of the undivided batch (outside the timed region)
    parity = None
    if not args.no_cpu_baseline:
        from oracle import c_oracle

        pat_id = {"gpt2_shaped": 0, "cl100k_shaped": 1, "o200k_shaped": 2, "o200k_custom8": 2}[args.encoding]
        C = c_oracle.COracle(pat_id, spec["mergeable_ranks"], spec["special_tokens"])
        ctoks, coff = C.encode_batch(blob, doc_off, None, ncpu)
        torch.cuda.set_device(devices[0])
        nd_all = len(doc_off) - 1
        g_off = torch.as_tensor(DevArray(do, nd_all + 1, "<i8"), device=f"cuda:{devices[0]}").cpu().numpy().astype(np.uint64)
        g_tok = torch.as_tensor(DevArray(dt, max(nt, 1), "<i4"), device=f"cuda:{devices[0]}")[:nt].cpu().numpy().view(np.uint32)
        parity = bool(np.array_equal(g_off, coff) and np.array_equal(g_tok, ctoks))
    print(json.dumps({"metric": "GB/s text encoded, one process driving N devices (host text in, ids gathered on device 0): PCIe-inclusive",
                      "value": round(base * args.steps / el / 1e9, 3), "unit": "GB/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
                      "ms_per_step": round(el / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
                      "data": "synthetic", "config": {"workload": f"{args.encoding}, {n} shards of {args.mib} MiB, tk_group_encode_batch_device",
                                                      "devices": devices, "tokens_total": int(nt),
                                                      "gather": "rccl" if core.group_stat("gathers_rccl") else "peer copies"},
                      "parity_all_tokens_vs_oracle": parity, "gather_verified": parity}), flush=True)
    if parity is False:
        print("bench: the gathered result differs from the oracle's", file=sys.stderr)
        sys.exit(3)


def generic_engine_figure(encoding: str, mib: int = 256):
    """Never `value`: the same pat_str through the generic engine (where any pat_str outside the three scanner families runs: the pattern
    as a DFA in LDS, tk_regex_dfa.inc) on the first `mib` MiB-sized corpus of the same generator, in a process of its own -- this very script
    with --generic-engine, every token compared with the oracle there.  A failure of that process is reported, not raised: the line of
    this run does not depend on it."""
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--generic-engine", "--mib", str(mib), "--steps", "3", "--warmup", "1", "--no-host-path", "--no-hf",
           "--cpu-sample-mib", "16", "--encoding", encoding]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
        lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
        j = json.loads(lines[-1])
        k = (j.get("roofline") or {}).get("kernels_ms_avg") or {}
        return {"gbps": j["value"], "ms_per_step": j["ms_per_step"], "mib": mib, "all_tokens_equal_to_the_oracle": j.get("parity_all_tokens_vs_oracle"),
                "kernels_ms_avg": {n: v for n, v in k.items() if "rx_" in n},
                "what": f"bench.py --generic-engine --mib {mib} --steps 3 --warmup 1 in a process of its own: the encoding's pat_str forced through the generic "
                        "regex engine (its DFA form) instead of the hand-written scanners, inputs resident in HBM, every token compared with the oracle"}
    except Exception as e:  # (time-out, no line, ...)
        return {"error": f"{type(e).__name__}: {str(e)[:200]}"}


def other_configs_figure():
    """BASELINE.json's other configurations at their full sizes -- C1 (gpt2-shaped, one 1 MiB document), C2 (cl100k-shaped, 64 MiB mixed
    UTF-8), C5 (o200k-shaped + 8 custom special tokens, 256 MiB, allowed_special="all") -- by tools/bench_configs.py in a process of its own:
    rate from HBM-resident inputs and every token against the oracle.  Reported beside the headline, never part of `value`."""
    import subprocess

    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_configs.py")], capture_output=True, text=True, timeout=400)
        out = {}
        for ln in r.stdout.strip().splitlines():
            if ln.startswith("{"):
                j = json.loads(ln)
                out[j["config"].split()[0]] = {k: j[k] for k in ("config", "bytes", "docs", "tokens", "ms_per_step", "GBps", "parity_all_tokens", "kernels_ms_avg") if k in j}
        return out or {"error": (r.stderr or "no output")[-200:]}
    except Exception as e:
        return {"error": f"{type(e).__name__}: {str(e)[:200]}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mib", type=int, default=1024, help="corpus size per GPU in MiB (headline: 1024)")
    ap.add_argument("--encoding", default="o200k_shaped")
    ap.add_argument("--cpu-sample-mib", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--parity-sample-mib", type=int, default=0,
                    help="N > 1: every rank compares the first this-many MiB of its shard with the oracle (0 = the whole shard, the default)")
    ap.add_argument("--no-host-path", action="store_true", help="skip the T2 / T3 host-boundary timings")
    ap.add_argument("--t3-sample-mib", type=int, default=64)
    ap.add_argument("--no-hf", action="store_true", help="skip the HF tokenizers context figure of the CPU baseline")
    ap.add_argument("--one-process", action="store_true",
                    help="NOT the driver's mode: drive the product's several-GPU entry (CoreBPE(devices=...), tk_group_encode_batch_device: host text in, "
                         "ids gathered on device 0 over xGMI) from this one process; a device is named several times when the box has fewer GPUs")
    ap.add_argument("--csrc-digest", action="store_true", help="print the digest of the kernel sources (what profiles/traffic.json is tied to) and exit")
    ap.add_argument("--generic-engine", action="store_true",
                    help="NOT the headline: run the encoding's pat_str on the generic regex engine instead of the hand-written scanners")
    args = ap.parse_args()
    if args.csrc_digest:
        print(csrc_digest())
        return

    import torch

    if args.one_process:
        return one_process(args, torch)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    # $TIKTOKEN_AMD_BENCH_BACKEND=gloo (a dry run of the N > 1 leg on a box with fewer GPUs than ranks: tools/gpu_bench_n2_dry.sh): the ranks
    # share the devices there are, collectives and the gather go through host copies over gloo.  Never the driver's mode, said in the line.
    backend = os.environ.get("TIKTOKEN_AMD_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    cdev = "cuda" if backend == "nccl" else "cpu"  # where the collectives' tensors live
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    import tiktoken_amd  # noqa: F401
    from tiktoken_amd.distributed import gather_tokens

    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    nbytes = args.mib << 20
    mix = 1  # web text
    seed = 0x5EED0003 if world == 1 else 0x5EED0004 + rank
    t0 = time.time()
    blob, doc_off = gen_corpus(seed, mix, nbytes, min(ncpu, 32))
    n_docs = len(doc_off) - 1
    t_gen = time.time() - t0

    # one process per GPU: the encoding's tables live on this rank's device
    from tiktoken_amd._tiktoken import CoreBPE
    from tiktoken_ext import amd_shaped

    spec = amd_shaped.ENCODING_CONSTRUCTORS[args.encoding]()
    if args.generic_engine:  # (read once, when the pattern is compiled)
        os.environ["TIKTOKEN_AMD_DEBUG"] = str(int(os.environ.get("TIKTOKEN_AMD_DEBUG", "0")) | 0x100000)
    core = CoreBPE(spec["mergeable_ranks"], spec["special_tokens"], spec["pat_str"], device=local_rank)

    d_text = torch.from_numpy(blob).cuda()          # nbytes + 64 readable bytes, resident in HBM
    d_off = torch.from_numpy(doc_off.view(np.int64)).cuda()
    torch.cuda.synchronize()

    # N > 1: rank 0 gathers every rank's token ids -- one grouped RCCL send / recv per step at the exact lengths (7 senders -> 7 xGMI
    # links into rank 0).  The transfer of step k runs while step k + 1 is being encoded: the library alternates between two pairs of
    # result buffers (tk_set_output_buffers), so the ids are sent from where the encoder left them, without a copy; drain() waits for
    # the last transfer INSIDE the timed region.  `value` includes the gather; `value_encode_only` (below) is the same loop without it.
    pending = [None]
    last_gather = [None]  # (per-rank tensors, counts) of the last completed gather (rank 0: what it RECEIVED)
    if world > 1:
        core.set_output_buffers(2)

    gather_mode = {"form": os.environ.get("TIKTOKEN_AMD_BENCH_GATHER", "exact")}  # "exact" | "padded" (the earlier form: a copy of the ids, one padded dist.gather)

    def step(gather=True):
        dt, nt, do = core.encode_batch_device(d_text.data_ptr(), nbytes, d_off.data_ptr(), doc_off, n_docs)
        if world > 1 and gather:
            toks = torch.as_tensor(DevArray(dt, max(nt, 1), "<i4"), device="cuda")  # (a view of the library's buffer: valid until the call after the next)
            if pending[0] is not None:
                last_gather[0] = pending[0].wait()
                torch.cuda.current_stream().synchronize()  # (wait() orders torch's stream only; the encoder runs on the library's: the buffer of step k - 1 is written again by step k + 1)
            if backend != "nccl":
                toks = toks[: max(nt, 1)].cpu()
            if gather_mode["form"] == "exact":
                pending[0] = gather_tokens(toks, nt, rank, world, dist, torch, async_op=True)
            else:
                pending[0] = gather_tokens(toks.clone(), nt, rank, world, dist, torch, async_op=True, padded=True)
        return dt, nt, do

    def drain():
        if pending[0] is not None:
            last_gather[0] = pending[0].wait()
            torch.cuda.current_stream().synchronize()
            pending[0] = None

    if world > 1 and gather_mode["form"] == "exact":
        # The form of the gather is agreed on by ALL ranks before anything is timed (a rank that fell back on its own would leave the
        # others inside a collective it never joins): one untimed exact-length exchange of a small buffer torch did not allocate --
        # the library's own result buffer, which is what a collective library may refuse -- then the failure flags are summed.
        # Two phases, so that a rank that fails cannot leave its peers inside an exchange it never joins: (1) everything LOCAL the exchange
        # needs (the encode, torch's view of the library's buffer, the host copy of the dry run) under try/except, the failure flags summed
        # by an all_reduce every rank reaches; (2) only if all ranks got that far, the exchange itself.  (A rank that raises INSIDE the
        # collective library leaves its peers to that library's own watchdog: nothing a caller can bound portably.)
        failed, probe, nt0 = 0, None, 0
        try:
            dt0, nt0, _ = core.encode_batch_device(d_text.data_ptr(), nbytes, d_off.data_ptr(), doc_off, n_docs)
            probe = torch.as_tensor(DevArray(dt0, max(nt0, 1), "<i4"), device="cuda")[: min(nt0, 1 << 16)]
            if backend != "nccl":
                probe = probe.cpu()
            torch.cuda.current_stream().synchronize()
        except Exception as e:
            print(f"bench: rank {rank} cannot hand the library's result buffer to torch ({type(e).__name__}: {str(e)[:120]})", file=sys.stderr)
            failed = 1
        flag = torch.tensor([failed], dtype=torch.int64, device=cdev)
        dist.all_reduce(flag)
        if int(flag.item()):
            gather_mode["form"] = f"padded (the library's buffer could not be wrapped on {int(flag.item())} rank(s))"
        else:
            try:
                gather_tokens(probe, int(probe.numel()) if nt0 else 0, rank, world, dist, torch, async_op=True).wait()
                torch.cuda.current_stream().synchronize()
            except Exception as e:
                print(f"bench: exact-length gather failed on rank {rank} ({type(e).__name__}: {str(e)[:120]})", file=sys.stderr)
                failed = 1
            flag = torch.tensor([failed], dtype=torch.int64, device=cdev)
            dist.all_reduce(flag)
            if int(flag.item()):
                gather_mode["form"] = f"padded (the exact-length exchange raised on {int(flag.item())} rank(s))"

    for _ in range(args.warmup):
        step()
    drain()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        dt, nt, do = step()
    drain()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        tot = torch.tensor([nbytes, nt], dtype=torch.int64, device=cdev)
        dist.all_reduce(tot)
        total_bytes, total_tokens = int(tot[0].item()), int(tot[1].item())
    else:
        total_bytes, total_tokens = nbytes, nt
    ms_per_step = elapsed / args.steps * 1e3
    value = total_bytes * args.steps / elapsed / 1e9
    value_encode_only = None
    # N > 1: the result of the last TIMED step -- this rank's ids and offsets, and on rank 0 what the gather delivered -- goes to host
    # memory now, before further steps write the (alternating) result buffers again; compared further down, outside every timed region
    own_result, gather_check = None, None
    if dist and not args.no_cpu_baseline:
        from tiktoken_amd.distributed import exchange_verdicts, ids_digest, verify_gathered

        own_result = (torch.as_tensor(DevArray(dt, max(nt, 1), "<i4"), device="cuda")[: nt].cpu().numpy().view(np.uint32),
                      torch.as_tensor(DevArray(do, n_docs + 1, "<i8"), device="cuda").cpu().numpy().astype(np.uint64))
        if rank == 0 and last_gather[0] is not None and last_gather[0][0] is not None:
            gather_check = [ids_digest(p.cpu().numpy()) for p in last_gather[0][0]]  # (count, digest) of what was RECEIVED, per rank
        last_gather[0] = None
    if dist:  # the same K steps without the gather: what the encoders do when nobody collects the ids on one rank
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(gather=False)
        torch.cuda.synchronize()
        dist.barrier()
        el2 = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=cdev)
        dist.all_reduce(el2, op=dist.ReduceOp.MAX)
        value_encode_only = round(total_bytes * args.steps / float(el2.item()) / 1e9, 3)

    # ---- per-kernel durations (HIP events on the library's stream), one profiled pass
    core.set_profiling(True)
    core.reset_kernel_ms()
    prof_steps = 2
    for _ in range(prof_steps):
        core.encode_batch_device(d_text.data_ptr(), nbytes, d_off.data_ptr(), doc_off, n_docs)
    core.set_profiling(False)
    kern = {}
    for k in KERNELS:
        ms, n = core.kernel_ms(k)
        if n:
            kern[k] = {"ms_total": ms, "launches": n, "ms_avg": ms / n}
    stats = core.last_stats()
    b_alg = nbytes + 4 * stats["tokens"] + 16 * (n_docs + 1)  # text in + u32 ids out + offsets in/out
    dom = max(kern, key=lambda k: kern[k]["ms_total"]) if kern else None
    sum_ms = sum(v["ms_total"] for v in kern.values()) / prof_steps if kern else None
    traffic, traffic_at, traffic_current = None, None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath) and dom:
        try:
            tj = json.load(open(tpath))
            if tj.get("workload_mib") == args.mib and tj.get("encoding") == args.encoding:
                traffic = tj.get("kernels", {}).get(dom, {}).get("hbm_bytes_per_launch")
                traffic_at = tj.get("measured_at_git")
                traffic_current = bool(tj.get("csrc_digest") == csrc_digest())
                if traffic is not None and not traffic_current and rank == 0:
                    print(f"bench: WARNING: profiles/traffic.json was measured on other kernel sources (csrc digest {tj.get('csrc_digest')}, "
                          f"git {traffic_at}) than this build ({csrc_digest()}): roofline.traffic is stale", file=sys.stderr)
        except Exception:
            traffic = None
    roofline = None
    if dom:
        achieved = b_alg / (kern[dom]["ms_avg"] * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                    "traffic_source": "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, not measured in this run)" if traffic else None,
                    "traffic_measured_at": traffic_at, "traffic_measured_on_these_kernel_sources": traffic_current,
                    "algorithmic_bytes_per_launch": b_alg, "kernel_ms_avg": round(kern[dom]["ms_avg"], 4),
                    "all_kernels_ms_per_step": round(sum_ms, 4),
                    "pipeline_achieved": round(b_alg / (ms_per_step * 1e-3) / 1e9, 2),  # whole pipeline: algorithmic bytes over the wall time of a step
                    "pipeline_frac": round(b_alg / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5),
                    "pipeline_achieved_over_summed_kernel_time": round(b_alg / (sum_ms * 1e-3) / 1e9, 2),  # (kernels overlap on side streams: a lower bound)
                    "kernels_ms_avg": {k: round(v["ms_avg"], 4) for k, v in kern.items()}}

    # ---- parity of the WHOLE result + CPU baseline (rank 0, N = 1 only)
    cpu = None
    parity = None
    gather_verified = None
    parity_detail = None
    if dist and own_result is not None:
        # Every rank compares ITS shard of the last timed step with the oracle (the host threads are shared between the ranks: ncpu / world
        # each; the whole shard unless --parity-sample-mib bounds it, the token count and a digest of the full id stream either way), the
        # verdicts travel in one all-gather, and rank 0 compares what it RECEIVED from each peer with what that peer says it sent.
        from oracle import c_oracle

        pat_id = {"gpt2_shaped": 0, "cl100k_shaped": 1, "o200k_shaped": 2, "o200k_custom8": 2}[args.encoding]
        C = c_oracle.COracle(pat_id, spec["mergeable_ranks"], spec["special_tokens"])
        g_toks, g_tok_off = own_result
        want = nbytes if args.parity_sample_mib <= 0 else min(nbytes, args.parity_sample_mib << 20)
        nd_s = n_docs if want >= nbytes else max(int(np.searchsorted(doc_off, want, side="right")) - 1, 1)
        sb = int(doc_off[nd_s])
        ctoks, coff = C.encode_batch(blob[:sb], doc_off[: nd_s + 1], None, max(1, ncpu // world))
        ok = bool(np.array_equal(g_tok_off[: nd_s + 1], coff) and np.array_equal(g_toks[: int(coff[-1])], ctoks))
        if not ok:
            print(f"PARITY MISMATCH on rank {rank} (seed {seed:#x}, first {nd_s} documents)", file=sys.stderr)
        cnt, dig = ids_digest(g_toks)
        verdicts = exchange_verdicts(cnt, dig, ok, rank, world, dist, torch, device=cdev)
        parity = bool(all(v[2] for v in verdicts))
        if rank == 0:
            parity_detail = {"documents_compared_on_rank0": nd_s, "bytes_compared_on_rank0": sb, "of_bytes": nbytes,
                             "per_rank_ok": [v[2] for v in verdicts], "tokens_per_rank": [v[0] for v in verdicts]}
            if gather_check is not None:
                per = [{"rank": r, "tokens_sent": v[0], "tokens_received": g[0], "digest_equal": bool(g == (v[0], v[1]))}
                       for r, (g, v) in enumerate(zip(gather_check, verdicts))]
                gather_verified = bool(len(per) == world and all(x["digest_equal"] for x in per))
                parity_detail["gather"] = per
            else:
                gather_verified = False
        del ctoks, coff, g_toks, g_tok_off
        own_result = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import c_oracle

        pat_id = {"gpt2_shaped": 0, "cl100k_shaped": 1, "o200k_shaped": 2, "o200k_custom8": 2}[args.encoding]
        C = c_oracle.COracle(pat_id, spec["mergeable_ranks"], spec["special_tokens"])
        # every document, every token of the timed workload against the oracle (all host threads)
        bufs = (np.empty(nbytes, np.uint32), np.empty(n_docs + 1, np.uint64))
        ctoks, coff = C.encode_batch(blob[:nbytes], doc_off, None, ncpu, out=bufs)  # (views of `bufs`: nothing below may write there)
        g_tok_off = torch.as_tensor(DevArray(do, n_docs + 1, "<i8"), device="cuda").cpu().numpy().astype(np.uint64)
        g_toks = torch.as_tensor(DevArray(dt, max(nt, 1), "<i4"), device="cuda")[: nt].cpu().numpy().view(np.uint32)
        parity = bool(np.array_equal(g_tok_off, coff) and np.array_equal(g_toks, ctoks))
        if not parity:
            bad_docs = np.flatnonzero(np.diff(g_tok_off.astype(np.int64)) != np.diff(coff.astype(np.int64)))
            d = int(bad_docs[0]) if len(bad_docs) else int(np.searchsorted(coff, int(np.flatnonzero(g_toks[: len(ctoks)] != ctoks[: len(g_toks)])[0]), side="right")) - 1
            print(f"PARITY MISMATCH: first differing document {d}: bytes [{int(doc_off[d])}, {int(doc_off[d + 1])}) "
                  f"{blob[int(doc_off[d]):int(doc_off[d]) + 120].tobytes()!r}", file=sys.stderr)
        # CPU baseline: the reference's scaling knob is one thread per document (core.py:175); the oracle's per-document phase is timed
        # inside the C code (its packing pass into one buffer is not part of the reference's work and is excluded), best of 3, on a bounded
        # sample; the single-thread rate on a smaller sample beside it
        # Thread sweep (the timed runs write into buffers of their OWN: `ctoks` / `coff` above are views of `bufs` and are compared
        # again below).  A thread count gets a sample in proportion (about 0.3 s of work each), capped at --cpu-sample-mib; the best
        # rate of the sweep is the baseline, `threads` the thread count that achieved it, `cores` what the box really gives (host_cores).
        sample_cap = min(nbytes, args.cpu_sample_mib << 20)
        tb = (np.empty(sample_cap + 64, np.uint32), np.empty(n_docs + 1, np.uint64))
        sweep = {}
        best_rate, best_thr, best_sb, best_nd = 0.0, 1, 0, 0
        for thr in sorted({t for t in (1, 8, 16, 32, 64, 128, 256) if t <= ncpu} | {ncpu if ncpu <= 256 else 256}):
            want = min(sample_cap, max(thr * (12 << 20), 16 << 20))
            nd_s = max(int(np.searchsorted(doc_off, want, side="right")) - 1, 1)
            sb = int(doc_off[nd_s])
            best = None
            for _ in range(3 if thr > 1 else 1):
                C.encode_batch(blob[:sb], doc_off[: nd_s + 1], None, thr, out=(tb[0], tb[1][: nd_s + 1]))
                el_c = c_oracle.last_encode_seconds()
                best = el_c if best is None else min(best, el_c)
            sweep[str(thr)] = round(sb / best / 1e9, 5)
            if sb / best / 1e9 > best_rate:
                best_rate, best_thr, best_sb, best_nd = sb / best / 1e9, thr, sb, nd_s
        cpu = {"value": round(best_rate, 4), "unit": "GB/s", "cores": host_cores(ncpu), "threads": best_thr, "kind": "port",
               "cores_what": "min(scheduler affinity, cgroup cpu.max quota) of this box; `threads` = the thread count of the sweep that gave the best rate",
               "single_thread_value": sweep.get("1"), "thread_sweep_gbps": sweep,
               "sample": f"first {best_nd} documents ({best_sb} bytes) of the same corpus, C restatement of CoreBPE (oracle/tk_oracle.c), one thread "
                         f"pool over documents (core.py:175), per-document encode phase only (timed in C), best of 3; sweep over thread counts "
                         f"with samples in proportion (12 MiB per thread, at most {args.cpu_sample_mib} MiB): the best rate is the baseline"}
        del tb

        if not args.no_hf:
            cpu["rust_cpu_tokenizer_for_context"] = hf_tokenizers_rate(args.encoding, spec, blob, doc_off, nbytes, ctoks, coff)

    # ---- host-boundary rates (rank 0, N = 1; never `value`): T2 = tk_encode_batch, host buffers in, host token ids out (pinned staging,
    # PCIe both ways inside the call); T3 = Encoding.encode_ordinary_batch on a bounded sample, Python list[str] -> list[list[int]]
    host_path = None
    if rank == 0 and world == 1 and not args.no_host_path:
        host_path = {}
        hb = blob[:nbytes]
        core.encode_batch_packed(hb, doc_off)
        best2, h_tok, h_off = None, None, None
        for _ in range(3):
            del h_tok, h_off  # (the result is a view of a page-locked buffer of the library's pool: given back before the next call, or that one pins a new GiB -- 60 ms)
            t0 = time.perf_counter()
            h_tok, h_off = core.encode_batch_packed(hb, doc_off)
            dt2 = time.perf_counter() - t0
            best2 = dt2 if best2 is None else min(best2, dt2)
        host_path["t2_gbps"] = round(nbytes / best2 / 1e9, 3)
        host_path["t2_ms"] = round(best2 * 1e3, 2)
        host_path["t2_what"] = ("tk_encode_batch: pageable host text + offsets in, token ids + offsets out in (page-locked) host memory, PCIe inclusive, best of 3; "
                                "the text goes by hipMemcpyAsync straight from the caller's pageable buffer in blocks of 64 MiB while chunks of 32 MiB are encoded "
                                "and their ids travel back on a second copy stream")
        host_path["t2_tokens"] = int(len(h_tok))
        # the link's rates on this box, measured now (a process of its own, 3 s), and T2 against what both directions at once allow: a GiB of
        # text goes in while about as many bytes of ids come out, so the bound of T2 is the duplex rate each way
        lr = link_rates()
        host_path["link"] = lr
        if lr.get("link_duplex_gbps_each_way"):
            host_path.update({"link_h2d_gbps": lr["link_h2d_gbps"], "link_d2h_gbps": lr["link_d2h_gbps"], "link_duplex_gbps": lr["link_duplex_gbps_each_way"],
                              "t2_over_duplex_bound": round(host_path["t2_gbps"] / lr["link_duplex_gbps_each_way"], 3)})
        if parity is not None:  # (g_toks / g_tok_off: the device-resident result that was compared with the oracle above)
            host_path["t2_identical_to_checked_result"] = bool(np.array_equal(h_off, g_tok_off) and np.array_equal(h_tok, g_toks))
        # decode (Encoding.decode_batch is one GPU call): token ids of the first 256 MiB of text in host memory -> bytes in host
        # memory (PCIe both ways inside the call), compared with the text they came from; kernel times from one more, profiled call
        ndd = max(int(np.searchsorted(doc_off, min(nbytes, 256 << 20), side="right")) - 1, 1)
        d_tok, d_off = h_tok[: int(h_off[ndd])], h_off[: ndd + 1]
        core.decode_batch_packed(d_tok[: 1 << 20], np.array([0, min(len(d_tok), 1 << 20)], np.uint64))
        dtd, first_ms = None, None
        for _ in range(3):  # (the first call of a size also page-locks its result buffer: reported on its own, the rate is the best of the three)
            d_bytes = None
            t0 = time.perf_counter()
            d_bytes, d_boff = core.decode_batch_packed(d_tok, d_off, as_array=True)  # (a uint8 view of the library's page-locked result buffer)
            el = time.perf_counter() - t0
            first_ms = round(el * 1e3, 2) if first_ms is None else first_ms
            dtd = el if dtd is None else min(dtd, el)
        core.set_profiling(True)
        core.reset_kernel_ms()
        core.decode_batch_packed(d_tok, d_off, as_array=True)
        core.set_profiling(False)
        host_path["decode_gbps"] = round(len(d_bytes) / dtd / 1e9, 3)
        host_path["decode_ms"] = round(dtd * 1e3, 2)
        host_path["decode_first_call_ms"] = first_ms
        host_path["decode_kernels_ms"] = {k: round(core.kernel_ms(k)[0], 4) for k in ("tk_k_dec_len", "tk_k_dec_copy")}
        host_path["decode_what"] = (f"tk_decode_batch: {len(d_tok)} token ids of the first {ndd} documents in host memory -> {len(d_bytes)} bytes + offsets in host "
                                    "memory (PCIe inclusive: the ids of one range travel in while the bytes of the range before travel out), best of 3; GB/s of decoded text")
        host_path["decode_identical_to_the_text"] = bool(np.array_equal(d_bytes, blob[: int(doc_off[ndd])]) and np.array_equal(d_boff, doc_off[: ndd + 1]))
        # the same ids resident in HBM, bytes left in HBM (tk_decode_batch_device): best of 3
        dd_tok = torch.from_numpy(np.array(d_tok).view(np.int32)).cuda()
        dd_off = torch.from_numpy(np.array(d_off).view(np.int64)).cuda()
        torch.cuda.synchronize()
        bestd = None
        for _ in range(3):
            t0 = time.perf_counter()
            dbp, dnb, dop = core.decode_batch_device(dd_tok.data_ptr(), len(d_tok), dd_off.data_ptr(), ndd)
            el = time.perf_counter() - t0
            bestd = el if bestd is None else min(bestd, el)
        host_path["decode_device_gbps"] = round(dnb / bestd / 1e9, 2)
        host_path["decode_device_ms"] = round(bestd * 1e3, 3)
        host_path["decode_device_identical_to_the_text"] = bool(
            dnb == int(doc_off[ndd]) and np.array_equal(torch.as_tensor(DevArray(dbp, max(dnb, 1), "|u1"), device="cuda")[:dnb].cpu().numpy(), blob[:dnb]))
        del dd_tok, dd_off
        del h_tok, h_off, d_tok, d_off, d_bytes
        nd3 = max(int(np.searchsorted(doc_off, min(nbytes, args.t3_sample_mib << 20), side="right")) - 1, 1)
        sb3 = int(doc_off[nd3])
        raw = blob[:sb3].tobytes()
        docs = [raw[int(doc_off[i]):int(doc_off[i + 1])].decode("utf-8") for i in range(nd3)]
        enc = tiktoken_amd.Encoding(args.encoding, pat_str=spec["pat_str"], mergeable_ranks=spec["mergeable_ranks"],
                                    special_tokens=spec["special_tokens"])
        enc.encode_ordinary_batch(docs[:64])
        t0 = time.perf_counter()
        lists = enc.encode_ordinary_batch(docs)
        dt3 = time.perf_counter() - t0
        host_path["t3_gbps"] = round(sb3 / dt3 / 1e9, 4)
        host_path["t3_ms"] = round(dt3 * 1e3, 1)
        host_path["t3_what"] = f"Encoding.encode_ordinary_batch(list[str]) -> list[list[int]] on the first {nd3} documents ({sb3} bytes), one run"
        del lists, docs, enc

    cold = None
    if rank == 0 and world == 1 and not args.no_host_path:
        # from a vocabulary FILE to an Encoding that can encode -- what get_encoding() costs after the library and the device
        # are up (file read + gunzip, native .tiktoken parser, table build on the host threads, upload); best of three
        from tiktoken_ext import amd_shaped
        ctor = amd_shaped.ENCODING_CONSTRUCTORS.get(args.encoding)
        if ctor is not None:
            best = None
            for _ in range(3):
                t0 = time.perf_counter()
                kw = ctor()
                t1 = time.perf_counter()
                e2 = tiktoken_amd.Encoding(**kw)
                t2 = time.perf_counter()
                e2.encode_ordinary("hello world")
                t3 = time.perf_counter()
                cur = {"cold_start_ms": round((t3 - t0) * 1e3, 1), "file_gunzip_parse_ms": round((t1 - t0) * 1e3, 1), "tables_and_upload_ms": round((t2 - t1) * 1e3, 1),
                       "first_call_ms": round((t3 - t2) * 1e3, 2)}
                if best is None or cur["cold_start_ms"] < best["cold_start_ms"]:
                    best = cur
                del e2, kw
            cold = dict(best, what=f"{args.encoding}: vocabulary file -> Encoding -> first encode(), library loaded and device initialised, best of 3")

    generic = None
    configs = None
    if rank == 0 and world == 1 and not args.no_host_path and not args.generic_engine:
        generic = generic_engine_figure(args.encoding)
        configs = other_configs_figure()

    if rank == 0:
        line = {
            "metric": "GB/s text encoded (o200k_base-shaped vocab, 1 GiB corpus per GPU), bit-exact vs CoreBPE restatement",
            "value": round(value, 3), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "value_encode_only": value_encode_only,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{args.encoding} encode_ordinary_batch, {args.mib} MiB synthetic web-text per GPU "
                                   f"(tkc_generate mix=1, seed {'0x5EED0003' if world == 1 else '0x5EED0004+rank'}), "
                                   f"{n_docs} docs on rank 0, inputs resident in HBM, packed u32 output",
                       "encoding": args.encoding, "pat_str_runs_on": "generic regex engine (--generic-engine)" if args.generic_engine else "hand-written scanners",
                       "bytes_per_gpu": nbytes, "docs_rank0": n_docs,
                       "rccl_ranks": (dist.get_world_size() if dist and backend == "nccl" else 0), "devices_visible": torch.cuda.device_count(),
                       "tokens_total": total_tokens, "pieces_rank0": stats["pieces"],
                       "parallelism": f"doc-sharded x{world}" + (f" + {'RCCL' if backend == 'nccl' else backend + ' (DRY RUN through host copies)'} gather of token ids to rank 0 ({gather_mode['form']} lengths)" if world > 1 else "")},
            "roofline": roofline, "cpu_baseline": cpu, "parity_all_tokens_vs_oracle": parity, "gather_verified": gather_verified,
            "parity_detail": parity_detail, "host_path": host_path,
            "vocabulary_tables": {"where": "HBM / L2: exact tables keyed by the piece's bytes (<= 8 bytes: the bytes; 9..23: 32-byte identity slots; longer: hash verified in the blob)",
                                  "lds_resident_hot_set": "measured in rounds 3-4 and not shipped: profiles/r04_lds_hot_set_closeout.txt", "front_workgroups_per_cu": core.stat("front_wgs_per_cu")},
            "cold_start": cold, "generic_engine": generic, "configs": configs,
            "host": {"cpus": ncpu, "nproc": os.cpu_count(), "cgroup_cpu_max": _read_first("/sys/fs/cgroup/cpu.max"),
                     "loadavg": _read_first("/proc/loadavg"), "corpus_gen_s": round(t_gen, 2)},
        }
        print(json.dumps(line), flush=True)
        hf = (cpu or {}).get("rust_cpu_tokenizer_for_context") or {}
        if parity is False or gather_verified is False or hf.get("same_ids_as_oracle_on_a_sample_of_documents") is False or \
                (host_path or {}).get("t2_identical_to_checked_result") is False or (host_path or {}).get("decode_identical_to_the_text") is False or \
                (host_path or {}).get("decode_device_identical_to_the_text") is False or \
                (generic or {}).get("all_tokens_equal_to_the_oracle") is False or \
                any(c.get("parity_all_tokens") is False for c in (configs or {}).values() if isinstance(c, dict)):
            print("bench: a parity check failed (see the line above)", file=sys.stderr)
            sys.exit(3)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
