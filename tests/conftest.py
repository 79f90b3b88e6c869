import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def synthetic_states():
    """Seed-0 synthetic PropagationNetwork / FusionNet state dicts (oracle/weights.py)."""
    import torch
    torch.set_grad_enabled(False)
    from oracle import weights as Wt
    return Wt.make_prop_state(0), Wt.make_fuse_state(0)


@pytest.fixture(scope="session", autouse=True)
def _bounded_cpu_threads():
    """The CPU oracle (oneDNN / OpenMP) scales poorly past a few dozen threads on these convolution sizes: on the 256-core
    host of the GPU box the default (one thread per core) is several times slower than 32 threads."""
    import torch
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    yield
