import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`gpu` tests need a device: without one they are skipped (a plain `pytest tests/` on a CPU box then runs the CPU
    tier and reports the GPU tier as skipped instead of failing with 'No HIP GPUs are available')."""
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs an MI355X (no GPU visible)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
