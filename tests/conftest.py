import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "eben_golden.npz"))


@pytest.fixture(scope="session")
def hip():
    """The loaded C-ABI library on a box with a GPU; GPU tests fail loudly if it is missing."""
    assert torch.cuda.is_available(), "GPU test selected but no HIP device is visible"
    from vibravox_amd import _lib

    return _lib.load()
