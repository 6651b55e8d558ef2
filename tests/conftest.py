import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True, scope="session")
def _bounded_cpu_threads():
    """The oracle is thousands of tiny PyTorch-CPU ops (T = 500 / 1000 serial GRU cell steps): on a many-core GPU host the
    default thread pool (one thread per core) turns every op into a barrier over 100+ threads.  8 threads is what the
    oracle scales to."""
    import torch
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(n, 8)))
    yield
