import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "pyramid-flow_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    # PF_TEST_VARIANT=<name>: run the suite against a measurement build of the library (pyramid-flow_amd/variants/<name>/,
    # `make -C pyramid-flow_amd/csrc variant NAME=...`) -- how a lab variant of a kernel earns its parity before it replaces
    # the shipping form.  Unset (the driver's runs): the shipping library.
    if os.environ.get("PF_TEST_VARIANT"):
        from pyflow_hip import lib
        lib.use_lab_library(os.environ["PF_TEST_VARIANT"])
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (dev container only)")


def pytest_collection_modifyitems(config, items):
    from oracle import shims
    if not shims.available():
        skip = pytest.mark.skip(reason="/root/reference not present (GPU box)")
        for it in items:
            if "reference" in it.keywords:
                it.add_marker(skip)


@pytest.fixture(autouse=True)
def _release_device_memory_between_tests(request):
    """GPU tests that spawn rank processes on the SAME device (tests/test_sp_gpu.py, test_bench_selflaunch_gpu.py: up to
    ~75 GiB per rank) need the memory this process's caching allocator still holds from earlier tests: hand it back to the
    driver after every GPU test (the full-depth / 768p tests leave > 200 GiB cached otherwise and a later rank runs out)."""
    yield
    if "gpu" in request.keywords:
        import gc
        import torch
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            gc.collect()
            torch.cuda.empty_cache()
