import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


_HIP_OPS = []


@pytest.fixture(scope="session")
def hip_ops():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ganslate_amd.hip.ops import HipOps
    _HIP_OPS.append(HipOps())
    return _HIP_OPS[0]


@pytest.fixture(autouse=True)
def _kernel_options_follow_the_environment():
    """tests flip GS_* switches with monkeypatch.setenv; the library's options (gs_set_option) are re-synchronised with
    the restored environment after every test (this fixture is set up before, hence torn down after, monkeypatch)"""
    yield
    for ops in _HIP_OPS:
        ops.sync_options()
