import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu():
    """Initialises the HIP library on cuda:0; GPU tests fail loudly (no fallback) when it is unavailable."""
    import torch
    from dart_amd import device
    assert torch.cuda.is_available(), "GPU test selected but no GPU is visible"
    device.init(0)
    return torch.device("cuda:0")
