"""pytest configuration: registers the `gpu` marker and puts the repo root, the product source root
(`sfmnext-impl_amd/`, laid out like the reference repo so `import layers, networks, trainer` resolve
to the MI355X build) and tests/golden on sys.path."""
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRODUCT = os.path.join(REPO, "sfmnext-impl_amd")
GOLDEN = os.path.join(REPO, "tests", "golden")
for p in (GOLDEN, REPO, PRODUCT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
    return load


def tt(a):
    return torch.from_numpy(np.ascontiguousarray(a))
