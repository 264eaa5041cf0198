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


def identity_residual_bound(M) -> float:
    """Bound for ||(L L^T) M - I||_F / sqrt(n) of an inverse factor L delivered in fp32 whose triangular inverse was
    accumulated in fp32 (csrc/invert.hip, round 5): the residual multiplies the forward error of L (~eps_fp32, below 1e-6)
    by ||M||, i.e. eps_fp32 * cond(M); 1e-4 where that is smaller (what an all-fp64 sweep rounded to fp32 meets)."""
    import torch
    w = torch.linalg.eigvalsh(M.double())
    return max(1e-4, 6e-8 * float(w[-1] / w[0]))
