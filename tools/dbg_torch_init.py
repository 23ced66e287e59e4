import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = {
 "A_core_then_cuda": "import torch; core=mk(); torch.zeros(1).cuda(); print('ok')",
 "B_cuda_then_core": "import torch; torch.zeros(1).cuda(); core=mk(); print('ok')",
 "C_core_encode_then_cuda": "import torch; core=mk(); core.encode_ordinary('hello world'); torch.zeros(1).cuda(); print('ok')",
 "D_devcount_then_cuda": "import torch; from tiktoken_amd import _lib; print(_lib.device_count()); torch.zeros(1).cuda(); print('ok')",
}
PRE = f"""
import sys; sys.path[:0]=[{ROOT!r}, {ROOT!r}+'/tests']
import helpers as h
def mk():
    from tiktoken_amd import CoreBPE
    g = h.load_golden('o200k_shaped')
    return CoreBPE(h.golden_vocab('o200k_shaped'), g['special_tokens'], g['pat_str'])
"""
for name, body in CASES.items():
    r = subprocess.run([sys.executable, "-c", PRE + body], capture_output=True, text=True, timeout=300)
    print(name, "rc", r.returncode, (r.stdout.strip().splitlines() or [''])[-1], "|", (r.stderr.strip().splitlines() or [''])[-1][:200], flush=True)
print(dict((k, v) for k, v in os.environ.items() if "VISIBLE" in k or "HSA" in k or "HIP" in k))
