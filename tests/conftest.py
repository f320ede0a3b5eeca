import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")
INPUTS = os.path.join(GOLD, "inputs")
for p in (REPO, os.path.join(REPO, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # On a GPU box PyTorch must initialise ITS HIP runtime before libhpf.so brings in /opt/rocm's: the other way round
    # torch.cuda.is_available() turns False for the rest of the process (the product does not need torch; the tests that check
    # "a GPU is present" and the RCCL gather do).
    if os.path.exists("/dev/kfd"):
        try:
            import torch
            torch.cuda.is_available()
        except Exception:
            pass


@pytest.fixture(scope="session")
def gold_dir():
    return GOLD


@pytest.fixture(scope="session")
def inputs_dir():
    return INPUTS
