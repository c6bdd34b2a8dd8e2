import os
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu via gpurun)')


@pytest.fixture(autouse=True)
def _scratch_cwd(tmp_path, monkeypatch):
    # ConfigValle() creates models/checkpoints + models/logs under the CWD (reference
    # behaviour, valle/config.py:74-77): keep that out of the repo.
    monkeypatch.chdir(tmp_path)
    yield


def has_gpu():
    import torch
    return torch.cuda.is_available()
