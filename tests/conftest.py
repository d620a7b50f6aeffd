import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A plain `pytest` on a machine without a GPU skips the gpu-marked tests instead of erroring in them (the product has no
    CPU path to fall back to).  On the GPU box nothing is skipped: a missing device there must fail loudly."""
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:                                   # noqa: BLE001
        have_gpu = False
    if have_gpu or os.environ.get("SSD_REQUIRE_GPU") == "1":
        return
    skip = pytest.mark.skip(reason="needs an MI355X (torch.cuda.is_available() is False)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
