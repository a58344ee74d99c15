import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "vae-time-series_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
    """-> dict of nested dicts of torch tensors from tests/golden/<name>.npz."""
    import numpy as np
    import torch

    blob = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    out = {}
    for key in blob.files:
        arr = blob[key]
        val = torch.from_numpy(arr.copy()) if arr.ndim > 0 else arr.item()
        node = out
        parts = key.split("/")
        # parameter names contain dots, never slashes
        for part in parts[:-1]:
            node = node.setdefault(part, {})
        node[parts[-1]] = val
    return out


@pytest.fixture(scope="session")
def golden():
    return load_golden
