#!/usr/bin/env python3
"""Can the RCCL branch of tk_group_encode_batch_device run on a ONE-GPU box?  It needs ncclCommInitAll over the group's devices; the
group's virtual ranks name device 0 twice.  This asks RCCL directly (ctypes, no torch): ncclCommInitAll(comms, 2, {0, 0})."""
import ctypes, os
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
lib = None
for name in ("librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"):
    try:
        lib = ctypes.CDLL(name)
        break
    except OSError:
        pass
assert lib is not None, "no RCCL"
lib.ncclGetErrorString.restype = ctypes.c_char_p
for devs in ([0], [0, 0]):
    comms = (ctypes.c_void_p * len(devs))()
    arr = (ctypes.c_int * len(devs))(*devs)
    rc = lib.ncclCommInitAll(comms, len(devs), arr)
    print(f"ncclCommInitAll(ndev={len(devs)}, devices={devs}) -> {rc} ({lib.ncclGetErrorString(rc).decode()})", flush=True)
    if rc == 0:
        for c in comms:
            lib.ncclCommDestroy(c)
