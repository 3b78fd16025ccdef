import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun / at round end)")
    config.addinivalue_line("markers", "slow: tens of seconds on the GPU box (large-sample statistics)")


def _has_gpu():
    return os.path.exists("/dev/kfd")


@pytest.fixture(scope="session")
def has_gpu():
    return _has_gpu()


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected with `-m gpu`; if they are collected on a box without a GPU
    # (plain `pytest tests/`), skip them instead of failing at hipInit.
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU (/dev/kfd absent)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
