import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The oracle legs of the parity tests are torch CPU code.  On the GPU box's 256-thread host torch's intra-op pool
    # oversubscribes the oracle's small products (G4's two oracle runs: 4 s on 8 threads, 45 s on 256); bench.py probes
    # the same effect for its cpu_baseline.  One cap for the whole session, before the first test runs.
    import torch
    if torch.get_num_threads() > 16:
        torch.set_num_threads(16)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
