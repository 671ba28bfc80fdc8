import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: CPU test that takes more than ~20 s")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_sessionstart(session):
    """The CPU oracle runs inside the GPU tests too: keep torch's intra-op threads to what this process may really use (cgroup CPU
    quota / affinity - a GPU box exposes all host cores but grants 16), or the oracle legs oversubscribe and take several times
    as long."""
    try:
        import torch
        import bench
        n, _, _ = bench.usable_cores()
        torch.set_num_threads(max(1, n))
    except Exception:       # noqa: BLE001  (never block the test session on this)
        pass
