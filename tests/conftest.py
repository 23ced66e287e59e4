import os
import sys

import pytest

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
