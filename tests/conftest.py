import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than ~20 s on 8 CPU threads")


@pytest.fixture(scope="session")
def lib():
    from endodav_amd import _lib

    return _lib.load()


@pytest.fixture(scope="session")
def cuda():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("test is marked gpu but no GPU is visible: the HIP path has no CPU fallback")
    return torch.device("cuda:0")
