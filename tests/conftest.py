import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "pyramid-flow_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (dev container only)")


def pytest_collection_modifyitems(config, items):
    from oracle import shims
    if not shims.available():
        skip = pytest.mark.skip(reason="/root/reference not present (GPU box)")
        for it in items:
            if "reference" in it.keywords:
                it.add_marker(skip)
