import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        # GPU box: make sure the in-tree HIP library matches the sources (no-op when it travelled up to date)
        from dsml_thesis_amd.build import build_lib
        build_lib(verbose=False)
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def rnd(seed, *shape):
    """Seeded standard-normal float32 tensor -- the same generator tools/make_golden.py uses."""
    return torch.from_numpy(np.random.RandomState(seed).standard_normal(shape).astype(np.float32))


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def gold():
    return golden
