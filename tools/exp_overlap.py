"""Experiment: do two independent encode pipelines overlap on one GPU (latency-bound kernels sharing CUs)?"""
import os, sys, time, threading, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import gen_corpus
from tiktoken_amd._tiktoken import CoreBPE
from tiktoken_ext import amd_shaped
spec = amd_shaped.o200k_shaped()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cores = [CoreBPE(spec["mergeable_ranks"], spec["special_tokens"], spec["pat_str"]) for _ in range(K)]
n = 1 << 30
blob, off = gen_corpus(0x5EED0003, 1, n, 32)
nd = len(off) - 1
d_text = torch.from_numpy(blob).cuda(); d_off = torch.from_numpy(off.view(np.int64)).cuda()
def one():
    cores[0].encode_batch_device(d_text.data_ptr(), n, d_off.data_ptr(), off, nd)
one(); torch.cuda.synchronize()
t0 = time.perf_counter(); [one() for _ in range(3)]; torch.cuda.synchronize(); t1 = (time.perf_counter() - t0) / 3
print("single pipeline: %.2f ms  %.1f GB/s" % (t1 * 1e3, n / t1 / 1e9))
# K pipelines, each 1/K of the documents (separate device copies so offsets start at 0)
cuts = [int(np.searchsorted(off, n * i // K)) for i in range(K)] + [nd]
parts = []
for i in range(K):
    a, b = cuts[i], cuts[i + 1]
    bo = int(off[a]); be = int(off[b])
    hb = np.zeros(be - bo + 64, np.uint8); hb[: be - bo] = blob[bo:be]
    ho = (off[a:b + 1] - off[a]).astype(np.uint64)
    parts.append((torch.from_numpy(hb).cuda(), torch.from_numpy(ho.view(np.int64)).cuda(), ho, be - bo, b - a))
def work(i):
    dt, do, ho, nb, ndd = parts[i]
    cores[i].encode_batch_device(dt.data_ptr(), nb, do.data_ptr(), ho, ndd)
for i in range(K): work(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    th = [threading.Thread(target=work, args=(i,)) for i in range(K)]
    [t.start() for t in th]; [t.join() for t in th]
torch.cuda.synchronize(); t2 = (time.perf_counter() - t0) / 3
print("%d concurrent pipelines: %.2f ms  %.1f GB/s" % (K, t2 * 1e3, n / t2 / 1e9))
