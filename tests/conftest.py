import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")
INPUTS = os.path.join(GOLD, "inputs")
for p in (REPO, os.path.join(REPO, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


# The test-suite is A/B tooling: its monkeypatched HPF_* switches reach hpf_create only with this opt-in (include/hpf.h, hpf_create_opts).
os.environ.setdefault("HPF_ENV_SWITCHES", "1")


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


def check_jacobian_checksums(J, g, rtol=1e-12):
    """the reference's own J0 of a synthetic feeder (oracle/make_golden.py run_case, full=False): nnz, J w, J^T w, row sums of |J|"""
    assert J.shape == tuple(g["J0_shape"])
    assert J.nnz == int(g["J0_nnz"]), (J.nnz, int(g["J0_nnz"]))
    w = np.cos(np.arange(J.shape[1]) * 0.37) + 1.5
    for ours, ref in ((J @ w, g["J0_matvec"]), (J.T @ w, g["J0_rmatvec"]), (np.asarray(abs(J).sum(axis=1)).ravel(), g["J0_absrowsum"])):
        assert np.abs(ours - ref).max() <= rtol * np.abs(ref).max()
