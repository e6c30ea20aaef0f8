import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_ops():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ganslate_amd.hip.ops import HipOps
    return HipOps()
