import os
import sys

import pytest

# PyTorch's ROCm wheels bundle a HIP runtime of their own, the product library links the system's: two runtimes in one process.  That works
# when torch's is loaded first; a process that initialises the system's first (the device count below, at collection) and imports torch
# afterwards finds "No HIP GPUs are available" in torch.  A run of the whole directory imports torch at collection anyway
# (test_distributed.py); a run of a single file gets the same order from here.
try:
    import torch  # noqa: F401
except ImportError:  # (the CPU-only tests do not need it)
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with `-m gpu`)")


@pytest.fixture(scope="session")
def have_gpu():
    import tiktoken_amd._lib as _lib

    return _lib.device_count() > 0


def pytest_collection_modifyitems(config, items):
    """`gpu` tests are skipped (not failed) on a machine without a HIP device; the driver's GPU box runs them."""
    import tiktoken_amd._lib as _lib

    if _lib.device_count() > 0:
        return
    if (config.getoption("markexpr", "") or "").strip() == "gpu":
        return  # `-m gpu` was asked for explicitly: fail loudly on a box without a device instead of reporting skips
    skip = pytest.mark.skip(reason="no HIP device visible (tiktoken_amd has no CPU path)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
