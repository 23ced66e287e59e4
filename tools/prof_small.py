"""Per-kernel HIP-event times of one small call (default: the C1 shape, 1 MiB Lorem ipsum, gpt2-shaped)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from tiktoken_amd._tiktoken import CoreBPE
from tiktoken_ext import amd_shaped
enc = sys.argv[1] if len(sys.argv) > 1 else "gpt2_shaped"
mib = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
spec = amd_shaped.ENCODING_CONSTRUCTORS[enc]()
core = CoreBPE(spec["mergeable_ranks"], spec["special_tokens"], spec["pat_str"])
words = "lorem ipsum dolor sit amet consectetur adipiscing elit sed do eiusmod tempor incididunt ut labore et dolore magna aliqua".split()
rng = np.random.default_rng(1)
n = int(mib * (1 << 20))
txt = (" ".join(rng.choice(words, size=n // 5)))[:n].encode()
host = np.zeros(len(txt) + 64, np.uint8); host[:len(txt)] = np.frombuffer(txt, np.uint8)
off = np.array([0, len(txt)], np.uint64)
d_text = torch.from_numpy(host).cuda(); d_off = torch.from_numpy(off.view(np.int64)).cuda()
def one(): return core.encode_batch_device(d_text.data_ptr(), len(txt), d_off.data_ptr(), off, 1)
for _ in range(3): one()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): one()
torch.cuda.synchronize(); print("ms per call: %.3f" % ((time.perf_counter() - t0) / 20 * 1e3))
core.set_profiling(True); core.reset_kernel_ms()
for _ in range(5): one()
core.set_profiling(False)
tot = 0
for k in bench.KERNELS:
    ms, cnt = core.kernel_ms(k)
    if cnt: print("  %-24s %.4f ms x %d" % (k, ms / cnt, cnt // 5)); tot += ms / 5
print("sum of kernels per call: %.3f ms" % tot)
