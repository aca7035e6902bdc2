import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def mmt_lib():
    """The product library; built on demand here, shipped prebuilt to the GPU box."""
    from mm_training_amd import build, _lib
    build.build()
    return _lib


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    class G:
        def __getitem__(self, name):
            return np.load(os.path.join(GOLDEN, name + ".npz"))
    return G()
