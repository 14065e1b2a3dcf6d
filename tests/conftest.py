import importlib
import os
import sys

import pytest

os.environ.setdefault("CRT_TEST_HOOKS", "1")   # the package then loads libcrt_hip_test.so: the product's objects + the unit-test hooks
# torch brings its own copy of the HIP runtime: when a test uses both torch.cuda and libcrt_hip.so in one process,
# torch's copy has to be the first one loaded (the other order leaves torch with "No HIP GPUs are available")
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG_NAME = "course-assignment-danielhalachev_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (its directory name has hyphens, so it is imported by string)."""
    return importlib.import_module(PKG_NAME)


@pytest.fixture(scope="session")
def scenes(pkg):
    return pkg.scenes


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_api
    oracle_api.lib()
    return oracle_api
