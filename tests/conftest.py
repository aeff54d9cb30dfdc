import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The one-pass kernel's float pass is taken for batches of 100 megapixels and more (ipx_runtime.hip, run_dev_any); the tests' batches
# are a few frames, so they ask for it whatever the size.  The float64 kernels are the "IPX_KS_FAST=0" variants of the same tests.
os.environ.setdefault("IPX_KS_FAST", "2")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
