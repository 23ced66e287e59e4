#!/usr/bin/env python3
"""Throughput + parity of the other BASELINE.json configs (run on the GPU box; bench.py covers the headline C3):

  C1  gpt2-shaped,  encode_ordinary on one 1 MiB ASCII Lorem-ipsum document
  C2  cl100k-shaped, encode_ordinary_batch on 64 MiB mixed UTF-8
  C5  o200k + 8 custom special tokens (encode_batch, allowed_special="all"), 256 MiB web text with one
      special per ~2 KiB plus decoys
  N1g the same generator as N1 at 1 GiB, the headline's size
  N1  (beside them, not one of BASELINE.json's) o200k-shaped on 256 MiB of text whose share of pieces that are not tokens is natural
      (tests/helpers.py natural_corpus: 2.3 %; the synthetic web text of C3 / C4 / C5: 18.7 %)

Each config is encoded from HBM-resident inputs (tk_encode_batch_device), timed over a few steps, and the
whole result is compared token-for-token with the CPU oracle.  Prints one JSON object per config.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch

    import helpers as h
    from bench import DevArray, KERNELS
    from tiktoken_amd._tiktoken import CoreBPE
    from tiktoken_ext import amd_shaped

    def run(name, enc_name, blob, off, allowed, steps=3):
        spec = amd_shaped.ENCODING_CONSTRUCTORS[enc_name]()
        core = CoreBPE(spec["mergeable_ranks"], spec["special_tokens"], spec["pat_str"])
        n = len(blob)
        host = np.zeros(n + 64, np.uint8)
        host[:n] = blob
        d_text = torch.from_numpy(host).cuda()
        d_off = torch.from_numpy(off.view(np.int64)).cuda()
        nd = len(off) - 1
        core.encode_batch_device(d_text.data_ptr(), n, d_off.data_ptr(), off, nd, allowed)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            dt, nt, do = core.encode_batch_device(d_text.data_ptr(), n, d_off.data_ptr(), off, nd, allowed)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / steps
        core.set_profiling(True)
        core.reset_kernel_ms()
        core.encode_batch_device(d_text.data_ptr(), n, d_off.data_ptr(), off, nd, allowed)
        core.set_profiling(False)
        kern = {k: round(core.kernel_ms(k)[0], 4) for k in KERNELS + ["tk_k_spec_cand", "tk_k_spec_resolve"] if core.kernel_ms(k)[1]}
        # parity: every document, every token
        pat_id = {"gpt2_shaped": 0, "cl100k_shaped": 1, "o200k_shaped": 2, "o200k_custom8": 2}[enc_name]
        C = h.c_oracle.COracle(pat_id, spec["mergeable_ranks"], spec["special_tokens"])
        nd_s = nd
        sb = int(off[nd_s])
        rt, ro = C.encode_batch(blob[:sb], off[: nd_s + 1], allowed, os.cpu_count() or 8)
        g_off = torch.as_tensor(DevArray(do, nd + 1, "<i8"), device="cuda")[: nd_s + 1].cpu().numpy().astype(np.uint64)
        g_tok = torch.as_tensor(DevArray(dt, max(nt, 1), "<i4"), device="cuda")[: int(g_off[-1])].cpu().numpy().view(np.uint32)
        ok = bool(np.array_equal(g_off, ro) and np.array_equal(g_tok, rt))
        print(json.dumps({"config": name, "encoding": enc_name, "bytes": n, "docs": nd, "tokens": nt, "ms_per_step": round(el * 1e3, 3),
                          "GBps": round(n / el / 1e9, 3), "parity_all_tokens": ok, "kernels_ms_avg": kern}), flush=True)

    only = set(sys.argv[1:])  # (python tools/bench_configs.py C2 C5: those configs only)
    for cfg, title, enc_name, steps in (("C1", "C1 gpt2 1MiB lorem, 1 doc", "gpt2_shaped", 10), ("C2", "C2 cl100k 64MiB mixed UTF-8", "cl100k_shaped", 3),
                                       ("C5", "C5 o200k+8 specials 256MiB, allowed_special=all", "o200k_custom8", 3),
                                       ("N1", "N1 o200k 256MiB of text with a natural miss rate (2.3 % of its pieces are not tokens; C3: 18.7 %): not a BASELINE configuration",
                                        "o200k_shaped", 3),
                                       ("N1g", "N1g the same text generator at the headline's size, 1 GiB (2.3 % of the pieces are not tokens; the headline corpus C3: 18.7 %): not a BASELINE configuration",
                                        "o200k_shaped", 3)):
        if only and cfg not in only:
            continue
        if cfg == "N1g" and "N1g" not in only and os.environ.get("TIKTOKEN_AMD_BENCH_N1G", "1") == "0":
            continue
        _, _, _, blob, off, allowed = h.baseline_config(cfg)
        run(title, enc_name, blob, off, allowed, steps=steps)


if __name__ == "__main__":
    main()
