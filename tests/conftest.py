import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_present():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # -m gpu on a box without a GPU must fail loudly, not skip: only guard the default run.
    if config.getoption("-m"):
        return
    if _gpu_present():
        return
    skip = pytest.mark.skip(reason="no GPU here; run with -m gpu on the MI355X box")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
