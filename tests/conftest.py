import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


class Golden:
    """Lazy view of a tests/golden/*.npz file; `.t(key)` returns a torch tensor."""

    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)

    def __getitem__(self, k):
        return self.z[k]

    def t(self, k):
        return torch.from_numpy(np.array(self.z[k]))

    def sd(self, prefix="sd/"):
        return {k[len(prefix):]: self.t(k) for k in self.z.files if k.startswith(prefix)}

    def keys(self):
        return self.z.files


def analytic_cotangent(shape, phase):
    """the fixed [B,C,H,W] cotangent tests/golden/make_golden.py used for its gradient fixtures (rebuilt, not stored)"""
    import numpy as np
    import torch
    B, C, H, W = shape
    b, c, y, x = np.meshgrid(np.arange(B), np.arange(C), np.arange(H), np.arange(W), indexing="ij")
    return torch.from_numpy(np.sin(0.37 * c + 0.11 * y + 0.05 * x + 1.3 * b + phase).astype(np.float32))


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]
    return get
