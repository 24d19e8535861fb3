import os
import sys

import pytest

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    """`gpu` tests need a ROCm GPU and the built HIP library: skip (not fail) without them."""
    import torch

    from pytsc_amd import _lib

    reason = None
    if not torch.cuda.is_available():
        reason = "no ROCm GPU in this container (run on the MI355X box: pytest -m gpu)"
    elif not os.path.exists(_lib.LIB_PATH):
        reason = f"{_lib.LIB_PATH} not built"
    if reason:
        skip = pytest.mark.skip(reason=reason)
        for item in items:
            if "gpu" in item.keywords:
                item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import pyoracle

    pyoracle.build()
    return pyoracle
