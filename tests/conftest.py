import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU spec-oracle (test infrastructure; see oracle/sad_oracle.c header)."""
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def sad():
    """The product package (import name of 3dsad-main_amd/)."""
    import sad_amd
    return sad_amd


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
