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
