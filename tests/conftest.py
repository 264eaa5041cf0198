import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def rel_fro(a, b):
    """Relative Frobenius error ||a - b|| / ||b|| in float64."""
    import torch
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    denom = torch.linalg.norm(b)
    return float(torch.linalg.norm(a - b) / denom) if denom > 0 else float(torch.linalg.norm(a - b))


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
